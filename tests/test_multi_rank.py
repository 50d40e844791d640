"""CPU, world_size 2 on gloo: the N>1 paths of bench.py.  Default mode: every rank codes its own tiles of the frame as independent pictures (here through the RD spine
over the CPU restatement, since the kernels need a GPU), the per-rank results are gathered and must equal the single-process runs, barrier + MAX reduction of the step time.
--kernels mode: CTU rows dealt round-robin to the ranks, every rank enumerating the PU jobs of its rows through the C ABI's host logic."""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hoputil import ROOT

sys.path.insert(0, ROOT)


def _free_port():
    """a port nobody listens on right now (a fixed one may still be held by an earlier run's sockets: the rendezvous would then wait for its timeout)"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def _worker(rank, world, port, W, H, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    hp = bench._hophip()
    L = hp.load()
    wctu, hctu = (W + 63) // 64, (H + 63) // 64
    rows = bench.rows_for_rank(hctu, world, rank)
    pred = (ctypes.c_int * 2)(0, -60); amvp = (ctypes.c_int * 4)(0, -60, -60, 0)
    jobs = np.zeros(425 * len(rows) * wctu, hp.PU_JOB_DTYPE)
    n = 0
    for r in rows:
        for c in range(wctu):
            n += L.hop_enumerate_ctu_jobs(W, H, r * wctu + c, 128, pred, 2, amvp, 498711, 3, 0, jobs.ctypes.data + n * jobs.itemsize, None, len(jobs) - n)
    jobs = jobs[:n]
    # every PU belongs to a CTU row of this rank
    assert set(np.unique(jobs["pu_y"] // 64).tolist()) <= set(rows)
    area = torch.tensor([float(np.sum(jobs["w"].astype(np.int64) * jobs["h"])), float(n), float(len(rows))], dtype=torch.float64)
    dist.all_reduce(area)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)      # stand-in for the per-rank step time
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put((area.tolist(), float(t.item())))
    dist.destroy_process_group()


def test_row_partition_world2():
    W, H = 456, 312            # ragged right/bottom CTUs (multiples of 8 like 7728x5368)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, W, H, q)) for r in range(2)]
    for p in procs:
        p.start()
    (area, n, nrows), tmax = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process enumeration of the whole frame for comparison
    import bench
    hp = bench._hophip()
    L = hp.load()
    wctu, hctu = (W + 63) // 64, (H + 63) // 64
    pred = (ctypes.c_int * 2)(0, -60); amvp = (ctypes.c_int * 4)(0, -60, -60, 0)
    tot_n = tot_area = 0
    for a in range(wctu * hctu):
        jobs = np.zeros(425, hp.PU_JOB_DTYPE)
        k = L.hop_enumerate_ctu_jobs(W, H, a, 128, pred, 2, amvp, 498711, 3, 0, jobs.ctypes.data, None, 425)
        tot_n += k
        tot_area += int(np.sum(jobs["w"][:k].astype(np.int64) * jobs["h"][:k]))
    assert (int(n), int(area), int(nrows)) == (tot_n, tot_area, hctu)
    assert tmax == 2.0
    # aligned part of the picture: every luma sample is covered 3 (part types) x 4 (depths) times; the ragged
    # right/bottom CTUs lose the depths whose CUs would cross the border (they are split, TEncCu.cpp:407-409)
    assert 12 * (W // 64 * 64) * (H // 64 * 64) < tot_area < 12 * W * H


def test_enumeration_interior_ctu():
    import bench
    hp = bench._hophip()
    L = hp.load()
    pred = (ctypes.c_int * 2)(0, -60); amvp = (ctypes.c_int * 4)(0, -60, -60, 0)
    jobs = np.zeros(2000, hp.PU_JOB_DTYPE); tags = np.zeros(2000, np.int32)
    k = L.hop_enumerate_ctu_jobs(7728, 5368, 121 * 5 + 7, 128, pred, 2, amvp, 498711, 3, 0, jobs.ctypes.data, tags.ctypes.data, 2000)
    assert k == 425                                   # 85 CUs x (2Nx2N + 2 Nx2N + 2 2NxN)
    ka = L.hop_enumerate_ctu_jobs(7728, 5368, 121 * 5 + 7, 128, pred, 2, amvp, 498711, 3, 1, jobs.ctypes.data, tags.ctypes.data, 2000)
    assert ka == 425 + 21 * 8                         # + 4 AMP shapes x 2 PUs for the 21 CUs >= 16
    shapes = set(zip(jobs["w"][:ka].tolist(), jobs["h"][:ka].tolist()))
    assert (12, 16) in shapes and (64, 48) in shapes and (4, 8) in shapes and (4, 16) in shapes
    # ragged bottom-right CTU of the 7728x5368 frame (48 x 56 samples): only CUs inside the picture
    kr = L.hop_enumerate_ctu_jobs(7728, 5368, 121 * 84 - 1, 128, pred, 2, amvp, 498711, 3, 0, jobs.ctypes.data, tags.ctypes.data, 2000)
    j = jobs[:kr]
    assert (j["pu_x"] + j["w"]).max() <= 7728 and (j["pu_y"] + j["h"]).max() <= 5368
    assert kr > 0 and int(j["w"].max()) <= 32          # the 64x64 CU crosses the border: split without a mode test


def _tile_worker(rank, world, port, FW, FH, tw, th, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from hoputil import lenslet
    from test_spine_cpu import run_cpu_wpp, spine_cpu
    Y, Cb, Cr = lenslet(FW, FH, 16, 11)
    (_, x, y), = bench.tiles_of_rank(FW, FH, tw, th, 1, rank)           # this rank's picture: a tile of the frame, as bench.py's default mode deals them out
    cost = run_cpu_wpp(spine_cpu(), tw, th, Y[y:y + th, x:x + tw], Cb[y // 2:(y + th) // 2, x // 2:(x + tw) // 2], Cr[y // 2:(y + th) // 2, x // 2:(x + tw) // 2], 5)[0]
    mine = torch.tensor(np.concatenate([[float(x), float(y)], cost]), dtype=torch.float64)
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)                                   # the merge of the per-rank results (bench.py itself only reduces the time)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put(([g.tolist() for g in got], float(t.item())))
    dist.destroy_process_group()


def test_independent_pictures_world2():
    """two ranks, each coding ITS tile of the frame as a picture (the default mode's sharding: no data-path collective); gathered costs == the single-process runs"""
    FW, FH, tw, th = 256, 64, 128, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tile_worker, args=(r, 2, port, FW, FH, tw, th, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, tmax = q.get(timeout=600)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from hoputil import lenslet
    from test_spine_cpu import run_cpu_wpp, spine_cpu
    Y, Cb, Cr = lenslet(FW, FH, 16, 11)
    assert sorted((int(g[0]), int(g[1])) for g in got) == [(0, 0), (128, 0)]
    for g in got:
        x, y = int(g[0]), int(g[1])
        cost = run_cpu_wpp(spine_cpu(), tw, th, Y[y:y + th, x:x + tw], Cb[y // 2:(y + th) // 2, x // 2:(x + tw) // 2], Cr[y // 2:(y + th) // 2, x // 2:(x + tw) // 2], 0)[0]
        assert g[2:] == cost.tolist(), (x, y)
    assert tmax == 2.0


def test_tiles_of_rank_partition_the_frame():
    """the default mode's split: every frame cut into whole tiles, numbered through the frames, `pictures` per rank, consecutive ranks taking consecutive pictures"""
    import bench
    fw, fh, tw, th, P = 7728, 5368, 1024, 256, 8
    seen = []
    for rank in range(8):
        t = bench.tiles_of_rank(fw, fh, tw, th, P, rank)
        assert len(t) == P
        for f, x, y in t:
            assert f == 0 and x % tw == 0 and y % th == 0 and x + tw <= fw and y + th <= fh
        seen += t
    assert len(set(seen)) == 64                       # 7 x 20 = 140 tiles in a frame: 8 ranks x 8 pictures are all different
    big = bench.tiles_of_rank(fw, fh, tw, th, 256, 1)  # 256 pictures per rank: rank 1 holds pictures 256..511 = frame 1 from tile 116 on, frame 2, frame 3 up to tile 91
    assert big[0] == (1, (116 % 7) * 1024, (116 // 7) * 256) and big[-1] == (3, (91 % 7) * 1024, (91 // 7) * 256) and len(set(big)) == 256


def test_bench_step_arithmetic_and_view_split():
    """bench.py's default mode: the ramp of a lag-5 wavefront (CTUs retired when the rows in flight first reach their maximum) and the round-robin split of BASELINE config
    5's 169 views over the ranks"""
    import bench
    done, rif = bench.wavefront_ramp_ctus(121, 84, 5)              # the 7728x5368 frame: 121 x 84 CTUs
    assert rif == 25 and done == sum(min(121, max(0, 120 - 5 * r)) for r in range(84))
    done2, rif2 = bench.wavefront_ramp_ctus(121, 2, 5)             # a two-row band: both rows run after 5 CTUs
    assert rif2 == 2 and done2 == 5
    seen = []
    for rank in range(8):
        v = bench.views_of_rank(169, 8, rank)
        assert all(x % 8 == rank for x in v) and len(v) in (21, 22)
        seen += v
    assert sorted(seen) == list(range(169))
    Y, Cb, Cr = bench.view_planes(624, 432, 3, 9)
    assert Y.shape == (432, 624) and Cb.shape == (216, 312) and Y.dtype == np.int16 and 0 <= Y.min() and Y.max() <= 255
    Y2 = bench.view_planes(624, 432, 4, 9)[0]
    assert not np.array_equal(Y, Y2)                                # neighbouring views differ (disparity + their own noise)


def _shard_worker(rank, world, port, W, H, seed, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib.util
    from hoputil import lenslet
    from test_spine_cpu import PART_DT, spine_cpu
    spec = importlib.util.spec_from_file_location("hop_shard", os.path.join(ROOT, "hevc-hop_amd", "shard.py"))
    sh = importlib.util.module_from_spec(spec); spec.loader.exec_module(sh)
    ag = sh.TorchAllgather(None)                                   # gloo: host tensors straight through
    L = spine_cpu()
    L.hop_spine_cpu_encode_shard_rank.restype = ctypes.c_long
    L.hop_spine_cpu_encode_shard_rank.argtypes = [ctypes.c_int] * 7 + [ctypes.c_void_p] * 8
    Y, Cb, Cr = [np.ascontiguousarray(p, np.int16) for p in lenslet(W, H, 16, seed)]
    n = ((W + 63) // 64) * ((H + 63) // 64)
    cost = np.zeros(n, np.float64); parts = np.zeros((n, 256), PART_DT); rec = np.zeros((H, W), np.int16)
    nc = L.hop_spine_cpu_encode_shard_rank(W, H, 32, 16, 5, rank, world, ctypes.cast(ag.fn, ctypes.c_void_p), None, Y.ctypes.data, Cb.ctypes.data, Cr.ctypes.data,
                                           cost.ctypes.data, parts.ctypes.data, rec.ctypes.data)
    ag.close()
    mine = torch.tensor(np.concatenate([[float(nc), float(ag.calls), float(np.sum(rec.astype(np.int64)))], cost]), dtype=torch.float64)
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    if rank == 0:
        q.put([g.tolist() for g in got])
    dist.destroy_process_group()


def test_one_picture_ctu_rows_over_two_ranks_gloo():
    """ONE picture's CTU rows dealt to two PROCESSES (rank g codes the rows r % 2 == g); after every wavefront step the finished CTUs travel through the all-gather
    callback of include/hophip.h's hop_encode_set_shard -- here hevc-hop_amd/shard.py's TorchAllgather on gloo, the spine over the CPU restatement (the kernels need a GPU; on
    the device the same spine code runs with RCCL behind the same callback, tests/test_gpu_spine.py).  Both ranks end with the whole picture, equal to the reference encoder's
    WaveFrontSynchro run (tests/golden/encoder_spine.npz)."""
    W, H, seed = 192, 128, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, W, H, seed, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=900)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    G = np.load(os.path.join(ROOT, "tests", "golden", "encoder_spine.npz"))
    want = G["%dx%d_seed%d_wpp/cost" % (W, H, seed)]
    assert got[0][1] == got[1][1] == 3 + 3 * 1                     # one exchange per wavefront step: cols + lag * (rows - 1) with lag = min(5, cols)
    assert got[0][0] >= 0 and got[1][0] >= 0 and got[0][2] == got[1][2]      # the same reconstruction on both ranks
    for g in got:
        assert np.array_equal(np.array(g[3:]), want)
