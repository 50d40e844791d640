"""The all-gather hop_encode_set_shard asks its caller for (include/hophip.h), on torch.distributed: RCCL (backend "nccl") between the GPUs of a node, gloo in the CPU
rehearsals.  One picture's CTU rows are dealt to the ranks (TEncSlice::compressSlice's loop under WaveFrontSynchro, TLibEncoder/TEncSlice.cpp:1027-1051); after
every wavefront step the host spine hands the step's finished CTUs -- about 25 KB each: reconstruction block, partition data, costs, coder states -- to this callback.

The callback is entered on one of the library's worker threads, on a fiber's small stack: it only passes the request to a Python thread of this object (which runs the
collective) and waits, so that no interpreter or torch frames pile up on that stack."""
import ctypes
import queue
import threading

import torch
import torch.distributed as dist

ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)


class TorchAllgather:
    """allgather = TorchAllgather(device) ; ctx.set_shard(rank, world, allgather) ; ... ; allgather.close().  device: the rank's torch.device for RCCL (payloads are staged
    through it), or None for a CPU (gloo) group."""

    def __init__(self, device=None, group=None):
        self.device, self.group = device, group
        self.world = dist.get_world_size(group)
        self.calls, self.bytes, self.error = 0, 0, None
        self._req, self._ans = queue.SimpleQueue(), queue.SimpleQueue()
        self._thread = threading.Thread(target=self._serve, daemon=True)
        self._thread.start()
        self.fn = ALLGATHER_FN(self._enter)          # (kept alive by this object)

    def _enter(self, user, send, recv, nbytes):
        self._req.put((send, recv, nbytes))
        return self._ans.get()

    def _serve(self):
        while True:
            r = self._req.get()
            if r is None:
                return
            send, recv, nbytes = r
            try:
                src = torch.frombuffer((ctypes.c_uint8 * nbytes).from_address(send), dtype=torch.uint8)
                dst = torch.frombuffer((ctypes.c_uint8 * (nbytes * self.world)).from_address(recv), dtype=torch.uint8)
                if self.device is None:
                    dist.all_gather_into_tensor(dst, src, group=self.group)
                else:
                    out = torch.empty(nbytes * self.world, dtype=torch.uint8, device=self.device)
                    dist.all_gather_into_tensor(out, src.to(self.device), group=self.group)
                    dst.copy_(out)                   # (synchronises)
                self.calls += 1
                self.bytes += nbytes * self.world
                self._ans.put(0)
            except Exception as e:                   # the library turns a non-zero return into HOP_ERR_STATE
                self.error = repr(e)
                self._ans.put(1)

    def close(self):
        self._req.put(None)
        self._thread.join(5)


class ThreadAllgather:
    """The same exchange between ranks that are THREADS of one process (tests: two contexts on one GPU): ThreadAllgather(world).rank(k) is rank k's callback object."""

    class _Rank:
        def __init__(self, owner, k):
            self.owner, self.k, self.calls = owner, k, 0
            self.fn = ALLGATHER_FN(self._enter)

        def _enter(self, user, send, recv, nbytes):
            o = self.owner
            try:
                o.slots[self.k] = ctypes.string_at(send, nbytes)
                o.barrier.wait(o.timeout)
                ctypes.memmove(recv, b"".join(o.slots), nbytes * o.world)
                o.barrier.wait(o.timeout)
                self.calls += 1
                return 0
            except threading.BrokenBarrierError:
                return 1

    def __init__(self, world, timeout=600.0):
        self.world, self.timeout = world, timeout
        self.slots = [b""] * world
        self.barrier = threading.Barrier(world)
        self.ranks = [ThreadAllgather._Rank(self, k) for k in range(world)]

    def rank(self, k):
        return self.ranks[k]
