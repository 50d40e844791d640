#!/usr/bin/env python3
"""bench.py -- CTUs/sec of the HEVC-HOP all-intra RD search on MI355X (one JSON line, see the driver contract).

DEFAULT (BASELINE.json's metric on BASELINE's configuration): the synthetic 7728x5368 lenslet frame (tests/hoputil.py:lenslet, micro-image pitch 15) coded as ONE
picture exactly as TEncSlice::compressSlice (TLibEncoder/TEncSlice.cpp:1000-1196) codes it with cfg/3DHencoder_intra_main.cfg --MIsize=15 and WaveFrontSynchro: for every
CTU every candidate TEncCu::xCompressCU tests -- merge / skip, SS + GT search for 2Nx2N, Nx2N, 2NxN and the AMP shapes with AMVP, merge and micro-image candidates, intra
2Nx2N / NxN with the 35-mode search and the transform tree, each with its residual quadtree, RDOQ and CABAC-counted bits --, the decision between them, the recursion over
CU sizes, the SS reference growing from the sentinel CU by CU, the coder contexts carried from CU to CU.  The candidates are evaluated by the HIP kernels of libhophip, the
decisions taken by the host spine (hevc-hop_amd/host/hop_spine.cpp); the CTU rows run as a lag-8 wavefront (at most 16 rows in flight) whose requests are served in batches.
The whole picture takes minutes, so it is coded CONTINUOUSLY on a thread of its own and a STEP is a fixed quantum of retired CTUs (one CTU row = 121) read from
hop_encode_progress: the wavefront's ramp and `--warmup` steps are untimed, `--steps` steps are timed, then the run is cancelled (hop_encode_cancel).  A wall-clock budget
(--budget-s, from process start) ends the timed region early if need be; the JSON says how many steps were timed.  `parity` in the JSON compares the RD costs of the CTUs
this very run retired with cost.csv of the unmodified reference encoder for the same picture (tests/golden/encoder_frame_mi15_rows*.npz, oracle/make_golden24.py).
`cpu_baseline` is that reference encoder (oracle/_ref/TAppEncoderRef) timed on a crop of the same frame; `roofline` / `kernels` come from a separate profiled pass with HIP
events around every launch.  If time is left, BASELINE's config 5 (169 sub-aperture views of 624x432 as independent pictures in one stacked context) is measured as a second,
separately named figure (`cfg5_views`).

Multi-GPU: `bench.py --gpus N` spawns N ranks itself (torch.distributed.run; or the driver does).  Default: every rank codes its own frame of the sequence (no data-path
collective, "scaling": "weak").  --shard-rows: ONE picture over all ranks -- the CTU rows dealt round-robin, every wavefront step's finished CTUs handed over by an all-gather
(hop_encode_set_shard, hevc-hop_amd/shard.py: RCCL between the GPUs of a node) --, "scaling": "strong".

--kernels: round 1's kernel-throughput mode (search kernels over a frozen, fully reconstructed SS reference: all CTUs independent).  It measures the kernels, not the
encode; its JSON says so.
"""
import argparse
import ctypes
import importlib.util
import json
import os
import sys
import time
PROC_T0 = time.time()

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

FRAME_W, FRAME_H, PITCH, QP = 7728, 5368, 15, 32
ALGO_BYTES_PER_CTU = 87040          # compulsory HBM traffic per CTU
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6      # MI355X vector FP64 (half the 157.3 TF FP32 vector rate of MI355X_MICROARCH.md)
INT_VALU_PEAK_TOPS = 78.6           # wave64 VALU issue: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (one v_sad_u16 = 2 abs-diff-acc)
WARP_FLOP_PER_SAMPLE = 24           # FP64 mul/add/sub of one warped sample (TComPrediction.cpp:925-968), conversions not counted


def _hophip():
    spec = importlib.util.spec_from_file_location("hophip", os.path.join(ROOT, "hevc-hop_amd", "hophip.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def lenslet_torch(W, H, pitch, seed, dev):
    """GPU version of tests/hoputil.py:lenslet (same structure; used only to make the big frame quickly)."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    yy = torch.arange(H, device=dev, dtype=torch.float32).view(H, 1)
    xx = torch.arange(W, device=dev, dtype=torch.float32).view(1, W)
    mx, my = torch.floor(xx / pitch), torch.floor(yy / pitch)
    ux, uy = xx - mx * pitch - pitch / 2.0, yy - my * pitch - pitch / 2.0
    sx, sy = mx * pitch * 0.12 + ux * 1.05, my * pitch * 0.12 + uy * 1.05

    def tex(ax, ay):
        t = None
        for _ in range(4):
            f = (torch.rand(2, generator=g) * 0.33 + 0.02).tolist()
            ph = (torch.rand(2, generator=g) * 6.283).tolist()
            a = float(torch.rand(1, generator=g) * 0.6 + 0.4)
            term = a * torch.sin(ax * f[0] + ph[0]) * torch.cos(ay * f[1] + ph[1])
            t = term if t is None else t + term
        return t / 4.0
    vign = torch.exp(-(ux ** 2 + uy ** 2) / (2 * (0.55 * pitch) ** 2))
    noise = torch.randn(H, W, generator=g).to(dev) * 2.0
    Y = torch.clamp(torch.round((0.5 + 0.45 * tex(sx, sy)) * vign * 255 + noise), 0, 255).to(torch.int16)
    cs = (slice(None, None, 2), slice(None, None, 2))
    sxc, syc, vc = sx.expand(H, W)[cs], sy.expand(H, W)[cs], vign[cs]
    Cb = torch.clamp(torch.round(128 + 0.25 * 255 * tex(sxc, syc) * vc), 0, 255).to(torch.int16)
    Cr = torch.clamp(torch.round(128 + 0.25 * 255 * tex(sxc, syc) * vc), 0, 255).to(torch.int16)
    return Y.contiguous(), Cb.contiguous(), Cr.contiguous()


def cu_rects(W, H):
    """legal CU rectangles covering the picture (quadtree, largest first) -- what xCopyYuv2SSRef commits"""
    out = []

    def rec(x, y, s):
        if x >= W or y >= H:
            return
        if x + s <= W and y + s <= H:
            out.append((x, y, s, 0))
        else:
            h = s // 2
            for q in range(4):
                rec(x + (q & 1) * h, y + (q >> 1) * h, h)
    for cy in range(0, H, 64):
        for cx in range(0, W, 64):
            rec(cx, cy, 64)
    return np.array(out, np.int32)


def rows_for_rank(hctu, world, rank):
    """CTU rows of one rank: round-robin (row r -> rank r % world), the interleaving SURVEY 8(e) proposes for the
    wavefront so that all ranks stay busy along the anti-diagonal."""
    return [r for r in range(hctu) if r % world == rank]


def gt_iters(w, h):
    m, it = min(w, h), 0
    while m > 1 and it < 6:
        it += 1
        m //= 2
    return it


def kernels_main(args):
    import torch
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE is %d: for N > 1 launch it as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` (one process per GPU)" % (args.gpus, world))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libhophip has no CPU path")
    # one process per GPU over RCCL; a box with fewer GPUs than ranks (rehearsal only) shares devices and falls back
    # to gloo for the barrier / MAX reduction -- the data path has no collective either way
    ndev = torch.cuda.device_count()
    shared = world > ndev
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    hp = _hophip()
    from hoputil import lambda_for_qp
    W, H = args.width, args.height
    wctu, hctu = (W + 63) // 64, (H + 63) // 64
    n_ctu = wctu * hctu
    lam, lc = lambda_for_qp(QP)

    # ---- pictures (resident before timing) ----
    Y, Cb, Cr = lenslet_torch(W, H, PITCH, 2, dev)
    gen = torch.Generator(device="cpu").manual_seed(7)
    recY = torch.clamp(Y + torch.randint(-2, 3, (H, W), generator=gen).to(dev).to(torch.int16), 0, 255).contiguous()
    ctx = hp.Context(W, H, device=local)
    L = ctx.L
    ctx.upload_orig(Y.cpu().numpy(), Cb.cpu().numpy(), Cr.cpu().numpy())
    rects = cu_rects(W, H)
    d_rects = torch.from_numpy(rects).to(dev)
    chk = ctx._chk
    chk(L.hop_ssref_commit_cus_device(ctx.h, len(rects), d_rects.data_ptr(), recY.data_ptr(), Cb.data_ptr(), Cr.data_ptr()), "commit")
    ctx.sync()

    # ---- PU job lists of this rank's CTU rows ----
    my_rows = rows_for_rank(hctu, world, rank)
    pred = (ctypes.c_int * 2)(0, -4 * PITCH)                     # one micro-image up
    amvp = (ctypes.c_int * 4)(0, -4 * PITCH, -4 * PITCH, 0)      # + one micro-image left
    flags = hp.HOP_FLAG_FEN | hp.HOP_FLAG_HADME
    L.hop_enumerate_ctu_jobs.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32,
                                                               ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    cap = 425 * len(my_rows) * wctu
    jobs = np.zeros(cap, hp.PU_JOB_DTYPE)
    tags = np.zeros(cap, np.int32)
    n = 0
    ctu_of_job = np.zeros(cap, np.int32)
    for r in my_rows:
        for c in range(wctu):
            k = L.hop_enumerate_ctu_jobs(W, H, r * wctu + c, 128, pred, 2, amvp, lc, flags, 0,
                                         jobs.ctypes.data + n * jobs.itemsize, tags.ctypes.data + n * 4, cap - n)
            assert 0 <= k <= cap - n
            ctu_of_job[n:n + k] = r * wctu + c
            n += k
    jobs, tags, ctu_of_job = jobs[:n], tags[:n], ctu_of_job[:n]
    my_ctus = len(my_rows) * wctu
    # 2Nx2N PUs (w == h == CU size) grouped by depth: the predictor runs depth by depth (no overlap inside a launch)
    is2n = (jobs["w"] == jobs["h"]) & (jobs["w"] == (64 >> (tags >> 16)))
    depth_idx = [np.nonzero(is2n & ((tags >> 16) == d))[0].astype(np.int32) for d in range(4)]
    d_jobs = torch.from_numpy(jobs.view(np.uint8)).to(dev)
    d_res = torch.zeros(n * hp.PU_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    d_idx = [torch.from_numpy(ix).to(dev) for ix in depth_idx]
    d_pj = [torch.zeros(max(1, len(ix)) * ctypes.sizeof(hp.PredJob), dtype=torch.uint8, device=dev) for ix in depth_idx]
    my_rects = rects[np.isin((rects[:, 1] // 64), my_rows)]
    d_myrects = torch.from_numpy(np.ascontiguousarray(my_rects)).to(dev)
    # ---- rows a7 / a9 / a10 / a12 of the same CUs: 35-mode intra rough search of every CU (+ the four 4x4 blocks of
    #      an 8x8 CU, NxN at maximum depth) ----
    cu = np.stack([jobs["pu_x"][is2n], jobs["pu_y"][is2n], jobs["w"][is2n]], axis=1).astype(np.int32)   # (x, y, size) of every CU of my CTU rows, all depths
    ij = []
    for S in (64, 32, 16, 8, 4):
        src = cu[cu[:, 2] == (8 if S == 4 else S)]
        if S == 4:                                                # the four 4x4 blocks of each 8x8 CU
            src = np.concatenate([src + np.array([dx, dy, 0], np.int32) for dy in (0, 4) for dx in (0, 4)])
        a = np.zeros(len(src), hp.INTRA_JOB_DTYPE)
        a["x"], a["y"], a["size"], a["strong"] = src[:, 0], src[:, 1], S, 1
        U = S // 4                                                # left column, corner and above row available inside the picture
        fl = np.zeros((len(src), 68), np.uint8)
        fl[:, U:2 * U] = (src[:, 0] > 0)[:, None]
        fl[:, 2 * U] = (src[:, 0] > 0) & (src[:, 1] > 0)
        fl[:, 2 * U + 1:3 * U + 1] = (src[:, 1] > 0)[:, None]
        a["flags"] = fl
        ij.append(a)
    intra_jobs = np.concatenate(ij)
    d_intra = torch.from_numpy(intra_jobs.view(np.uint8)).to(dev)
    d_satd = torch.zeros(len(intra_jobs) * 35, dtype=torch.int32, device=dev)
    # the residual quadtree leaf (row a8b + a11 + the CABAC counter) of the same CUs: forward transform, estBit, RDOQ, counted bits,
    # inverse path, SSE and the cbf-zero decision of every component TU, all from the slice's initial ISS context snapshot
    LAM = 0.57 * 2.0 ** ((QP - 12) / 3.0)
    CW = 2.0 ** ((QP - (QP - 1)) / 3.0)                           # chroma distortion weight for the chroma QP the table gives at QP 32
    snap = np.zeros((1, hp.CABAC_CTX_BYTES), np.uint8)
    L.hop_cabac_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    chk(L.hop_cabac_init(snap.ctypes.data, 3, QP), "cabac_init")
    d_snap = torch.from_numpy(snap).to(dev)
    tu_by_depth = []
    for d in range(4):
        S = 64 >> d
        src = cu[cu[:, 2] == S]
        parts = []

        def tus(comp, N, T, qp, lamq, w):
            for oy in range(0, N, T):
                for ox in range(0, N, T):
                    a = np.zeros(len(src), hp.TU_RD_JOB_DTYPE)
                    m = 2 if comp else 1
                    a["x"], a["y"], a["comp"], a["log2_size"], a["qp_scaled"] = src[:, 0] + m * ox, src[:, 1] + m * oy, comp, T.bit_length() - 1, qp
                    a["tr_depth"], a["sign_hide"], a["bit_depth"] = (1 if N > T else 0), 1, 8
                    a["lambda_rdoq"], a["lambda_rd"], a["dist_weight"] = lamq, LAM, w
                    parts.append(a)
        tus(0, S, min(S, 32), QP, LAM, 1.0)                       # luma TUs: the CU, or four 32x32 for a 64x64 CU
        C = max(S // 2, 4)                                        # chroma TUs (an 8x8 CU codes one 4x4 per plane)
        for comp in (1, 2):
            tus(comp, C, min(C, 16), QP - 1, LAM / CW, CW)      # a 64x64 CU splits into 32x32 luma + 16x16 chroma TUs
        tu_by_depth.append(np.concatenate(parts))
    d_tu = [torch.from_numpy(t.view(np.uint8)).to(dev) for t in tu_by_depth]
    d_tur = [torch.zeros(len(t) * hp.TU_RD_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev) for t in tu_by_depth]
    tu_off = [np.concatenate([[0], np.cumsum(1 << (2 * t["log2_size"].astype(np.int64)))]) for t in tu_by_depth]
    d_tuoff = [torch.from_numpy(o[:-1].copy()).to(dev) for o in tu_off]
    d_levels = torch.zeros(int(max(o[-1] for o in tu_off)), dtype=torch.int32, device=dev)
    L.hop_tu_rd_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    for c3, pl in enumerate((recY, Cb, Cr)):                      # neighbours of the intra search: the reconstruction = the frozen reference
        host_plane = np.ascontiguousarray(pl.cpu().numpy(), np.int16)
        chk(L.hop_recon_upload(ctx.h, c3, host_plane.ctypes.data), "recon_upload")
    # --rqt: the full transform-size search of the same CUs (three levels of transform units, transform-skip retry, context chaining, recount)
    if args.rqt:
        d_rq, d_rqr, rq_cls, d_rqc, d_rqx, d_rqf, d_rqs, d_rqb = [], [], [], [], [], [], [], []
        cu_snap = np.zeros((1, hp.CABAC_CU_CTX_BYTES), np.uint8)
        L.hop_cabac_cu_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        chk(L.hop_cabac_cu_init(cu_snap.ctypes.data, 3, QP), "cabac_cu_init")
        d_cusnap = torch.from_numpy(cu_snap).to(dev)
        for d in range(4):
            S = 64 >> d
            src = cu[cu[:, 2] == S]
            a = np.zeros(len(src), hp.RQT_JOB_DTYPE)
            a["x"], a["y"], a["log2_cu"], a["ctx_index"] = src[:, 0], src[:, 1], 6 - d, 0
            a["qp_scaled"] = (QP, QP - 1, QP - 1); a["sign_hide"] = 1; a["use_ts"] = 1; a["log2_max_tu"] = 5; a["log2_min_tu_in_cu"] = (4, 3, 2, 2)[d]
            a["lambda_rd"] = LAM; a["lambda_rdoq"] = (LAM, LAM / CW, LAM / CW); a["dist_weight"] = (CW, CW)
            rq_cls.append(a[:1].copy())
            d_rq.append(torch.from_numpy(a.view(np.uint8)).to(dev))
            d_rqr.append(torch.zeros(len(a) * hp.RQT_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev))
            d_rqc.append(torch.zeros(len(a) * S * S * 3 // 2, dtype=torch.int32, device=dev))            # chosen levels
            d_rqx.append(torch.zeros(len(a) * hp.CABAC_CTX_BYTES, dtype=torch.uint8, device=dev))       # coder state after the quadtree
            d_rqf.append(torch.zeros(len(a) * 4, dtype=torch.int32, device=dev))                        # root cbf + final distortions
            sy = np.zeros(len(a), hp.CU_SYNTAX_DTYPE)                                                   # synthetic CU syntax: 2Nx2N, no merge, an MVD, GT flag with small vectors
            sy["n_pu"], sy["max_merge_cand"], sy["amp_acc"], sy["is_min_cu"] = 1, 5, int(S >= 16), int(S == 8)
            sy["pu"]["mvd"][:, 0] = (-17, 5); sy["pu"]["gt_flag"][:, 0] = 1; sy["pu"]["gt"][:, 0] = (1, 0, -1, 2, 0, 1, 0, 0)
            d_rqs.append(torch.from_numpy(sy.view(np.uint8)).to(dev))
            d_rqb.append(torch.zeros(len(a) * 2, dtype=torch.int32, device=dev))                        # bits, skipped
        L.hop_rqt_device.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 6
        L.hop_rqt_finish_device.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 6
        L.hop_inter_cu_bits_device.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 11
    CH = int(os.environ.get("HOP_BENCH_CH", max(n, 1)))               # PUs per hop_me_search_device call: the whole frame (the library cuts it into stream lanes); 128 k chunks cost 6 %
    jsz, rsz = hp.PU_JOB_DTYPE.itemsize, hp.PU_RESULT_DTYPE.itemsize

    def step():
        for o in range(0, n, CH):
            m = min(CH, n - o)
            chk(L.hop_me_search_device(ctx.h, m, d_jobs.data_ptr() + o * jsz, d_res.data_ptr() + o * rsz, hp.HOP_STAGE_GT), "me_search")
        for d in range(4):
            k = len(depth_idx[d])
            if k:
                chk(L.hop_pred_jobs_from_results_device(ctx.h, k, d_idx[d].data_ptr(), d_jobs.data_ptr(), d_res.data_ptr(), d_pj[d].data_ptr()), "pred_jobs")
                chk(L.hop_pred_inter_device(ctx.h, k, d_pj[d].data_ptr()), "pred")
                if args.rqt:
                    nq = int(d_rq[d].numel() // hp.RQT_JOB_DTYPE.itemsize)
                    chk(L.hop_rqt_device(ctx.h, nq, d_rq[d].data_ptr(), rq_cls[d].ctypes.data, d_snap.data_ptr(), d_rqr[d].data_ptr(), d_rqc[d].data_ptr(), d_rqx[d].data_ptr()), "rqt")
                    # root-cbf-zero test, reconstruction into the context's picture, final distortions (tail of encodeResAndCalcRdInterCU)
                    chk(L.hop_rqt_finish_device(ctx.h, nq, d_rq[d].data_ptr(), rq_cls[d].ctypes.data, d_rqr[d].data_ptr(), d_rqc[d].data_ptr(), d_rqx[d].data_ptr(), d_rqf[d].data_ptr()), "rqt_finish")
                    # the CU-level syntax bits from the slice's initial coder state (xAddSymbolBitsInter)
                    chk(L.hop_inter_cu_bits_device(ctx.h, nq, d_rq[d].data_ptr(), rq_cls[d].ctypes.data, d_rqs[d].data_ptr(), d_rqr[d].data_ptr(), d_rqc[d].data_ptr(), d_snap.data_ptr(),
                                                   d_cusnap.data_ptr(), d_rqb[d].data_ptr(), d_rqb[d].data_ptr() + 4 * nq, None, None), "cu_bits")
                else:
                    chk(L.hop_tu_rd_device(ctx.h, len(tu_by_depth[d]), d_tu[d].data_ptr(), d_snap.data_ptr(), d_tuoff[d].data_ptr(), int(tu_off[d][-1]),
                                           d_levels.data_ptr(), d_tur[d].data_ptr()), "tu_rd")
        chk(L.hop_intra_rough_device(ctx.h, len(intra_jobs), d_intra.data_ptr(), d_satd.data_ptr()), "intra_rough")
        chk(L.hop_ssref_commit_cus_device(ctx.h, len(my_rects), d_myrects.data_ptr(), recY.data_ptr(), Cb.data_ptr(), Cr.data_ptr()), "commit")

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    L.hop_profile_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.hop_profile_reset.argtypes = [ctypes.c_void_p]
    L.hop_profile_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.hop_set_lanes.argtypes = [ctypes.c_void_p, ctypes.c_int]
    # ---- the timed region: exactly `steps` steps, profiling OFF (no event pairs around the launches) ----
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # ---- a separate, untimed profiling pass for the roofline: ONE stream lane, so that no two kernels share the device and the
    #      HIP-event durations (recorded on the library's stream) are exclusive.  profiles/r02_bench_excl_kernel_stats.csv is the
    #      rocprofv3 --kernel-trace --stats view of the same pass (HOP_LANES=1). ----
    prof_steps = max(1, min(2, args.steps))
    chk(L.hop_set_lanes(ctx.h, 1), "set_lanes")
    chk(L.hop_profile_reset(ctx.h), "profile_reset")
    chk(L.hop_profile_enable(ctx.h, 1), "profile_enable")
    barrier()
    tp0 = time.perf_counter()
    for _ in range(prof_steps):
        step()
    barrier()
    prof_ms_per_step = (time.perf_counter() - tp0) / prof_steps * 1e3
    prof = {}
    names = {0: "k_ss_search", 1: "k_frac", 2: "k_gt_search", 3: "k_pred_inter", 4: "k_ssref_commit", 6: "k_tu_rd (transform + setup + inverse + decide)", 7: "k_intra_rough", 8: "k_rdoq", 9: "k_coeff_bits"}
    for kid, name in names.items():
        la, ms, un = ctypes.c_uint64(), ctypes.c_double(), ctypes.c_uint64()
        chk(L.hop_profile_read(ctx.h, kid, ctypes.byref(la), ctypes.byref(ms), ctypes.byref(un)), "profile_read")
        prof[name] = {"launches": la.value, "total_ms": ms.value, "units": un.value, "ms_per_step": ms.value / prof_steps}
    chk(L.hop_profile_enable(ctx.h, 0), "profile_disable")
    # a result checksum so that a run can be compared with another build
    res_host = np.frombuffer(d_res.cpu().numpy().tobytes(), hp.PU_RESULT_DTYPE)

    tu_crc = 0
    for d in range(4):
        if args.rqt:
            rr = np.frombuffer(d_rqr[d].cpu().numpy().tobytes(), hp.RQT_RESULT_DTYPE)
            ff = d_rqf[d].cpu().numpy().reshape(-1, 4).astype(np.uint64)
            ff[:, 0] += d_rqb[d].cpu().numpy()[:len(ff)].astype(np.uint64) * 5
            tu_crc ^= int(np.bitwise_xor.reduce((rr["bits"].astype(np.uint64) * 31 + rr["dist"] + rr["tr_idx"][:, 0].astype(np.uint64) * 7 + ff[:, 0] * 3 + ff[:, 1] + ff[:, 2] + ff[:, 3]) *
                                                np.arange(1, len(rr) + 1, dtype=np.uint64)) & np.uint64(0xFFFFFFFF))
            continue
        tr = np.frombuffer(d_tur[d].cpu().numpy().tobytes(), hp.TU_RD_RESULT_DTYPE)
        tu_crc ^= int(np.bitwise_xor.reduce((tr["bits"].astype(np.uint64) * 31 + tr["dist"] + tr["abs_sum"] * 7) * np.arange(1, len(tr) + 1, dtype=np.uint64)) & np.uint64(0xFFFFFFFF))

    if rank == 0:
        value = n_ctu * args.steps / dt
        dom = max(prof, key=lambda k: prof[k]["total_ms"])
        p = prof[dom]
        avg_ms = p["total_ms"] / max(1, p["launches"])
        ctus_per_launch = my_ctus * prof_steps / max(1, p["launches"])
        achieved = ALGO_BYTES_PER_CTU * ctus_per_launch / (avg_ms * 1e-3) / 1e9
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
        # (profiles/r01_traffic.json; PMC counters cannot be read from inside the process)
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))["kernels"].get(dom)
            if tj and W == FRAME_W and H == FRAME_H and world == 1:
                traffic = tj["hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            traffic = None
        # honest VALU-side figures (neither HBM nor MFMA binds this path)
        st = 1 + 2
        samples = float(np.sum(st * np.array([gt_iters(int(w), int(h)) for w, h in zip(jobs["w"], jobs["h"])]) * 56.0 * jobs["w"] * jobs["h"]))
        gt_tflops = samples * WARP_FLOP_PER_SAMPLE * prof_steps / (prof["k_gt_search"]["total_ms"] * 1e-3) / 1e12 if prof["k_gt_search"]["total_ms"] else 0.0
        win = (jobs["rng_right"] - jobs["rng_left"] + 1).clip(0) * (jobs["rng_bottom"] - jobs["rng_top"] + 1).clip(0)
        sad_ops = float(np.sum(win.astype(np.float64) * jobs["w"] * jobs["h"] / np.where(jobs["h"] > 8, 2, 1) / 2.0))   # v_sad_u16 lane-ops
        ss_tops = sad_ops * prof_steps / (prof["k_ss_search"]["total_ms"] * 1e-3) / 1e12 if prof["k_ss_search"]["total_ms"] else 0.0
        out = {
            "metric": "kernel throughput only (NOT the encode metric): CTUs/sec of the search kernels over a frozen, fully reconstructed SS reference, 7728x5368 lenslet @QP32",
            "value": value, "unit": "CTU/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "i16+f64", "data": "synthetic",
            "config": {"workload": "synthetic lenslet %dx%d pitch %d, QP%d, HOP on: SS+-128 (FEN) + frac (HAD) + GT search + GT predictor + %s + 35-mode intra rough search + SS-ref commit, "
                                   "full symmetric RD-tree PU set (%d PUs/frame), frozen SS reference, no host RD spine" % (W, H, PITCH, QP,
                                   "the whole residual-quadtree search of its residual (three transform sizes, transform-skip retry, context chaining, recount; Y,Cb,Cr) + root-cbf test, reconstruction, final distortion and the CU-level syntax bits = encodeResAndCalcRdInterCU of the candidate" if args.rqt else
                                   "residual-quadtree leaf of its residual (DCT, RDOQ, CABAC-counted bits, inverse, SSE, cbf decision; Y,Cb,Cr)", n if world == 1 else -1),
                       "ctus": n_ctu, "pus_rank0": int(n), "parallelism": "ctu-rows-rr%d" % world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": avg_ms, "ctus_per_launch": ctus_per_launch, "algorithmic_bytes_per_ctu": ALGO_BYTES_PER_CTU,
                         "note": "VALU-bound path: see valu_fp64 / valu_int for the binding rooflines"},
            "valu_fp64": {"kernel": "k_gt_search", "achieved": gt_tflops, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": gt_tflops / FP64_VECTOR_PEAK_TFLOPS,
                          "flop_per_warped_sample": WARP_FLOP_PER_SAMPLE, "warped_samples_per_s": gt_tflops * 1e12 / WARP_FLOP_PER_SAMPLE,
                          "note": "reference-equivalent work (24 double flops per warped sample in the reference's formulation); the kernel evaluates the "
                                  "same warp in exact integer arithmetic, about 7 double-rate + 30 integer VALU operations per sample"},
            "valu_int": {"kernel": "k_ss_search", "achieved": ss_tops, "peak": INT_VALU_PEAK_TOPS, "unit": "T v_sad_u16 lane-op/s", "frac": ss_tops / INT_VALU_PEAK_TOPS,
                         "note": "reference-equivalent work (one full search per PU); CU families derive the five symmetric PUs of a CU from one set of "
                                 "quadrant SADs, so about a third of these v_sad_u16 are executed"},
            "kernels": prof,
            "kernels_note": "HIP-event time per launch from a separate untimed pass of %d step(s) on ONE stream lane (hop_set_lanes 1): exclusive durations; "
                            "that pass ran at %.1f ms per step, the timed region (2 lanes, profiling off) at ms_per_step" % (prof_steps, prof_ms_per_step),
            "tu_rd_crc": tu_crc,
            "result_crc": int(np.bitwise_xor.reduce(res_host["cost"].astype(np.uint64) * np.arange(1, len(res_host) + 1, dtype=np.uint64)) & np.uint64(0xFFFFFFFF)),
        }
        if world == 1:
            out["cpu_baseline"] = cpu_baseline(hp, ctx, jobs, ctu_of_job, is2n, res_host, Y, W, H, wctu, lc, args.cpu_ctus)
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(hp, ctx, jobs, ctu_of_job, is2n, res_gpu, Y, W, H, wctu, lc, n_ctus):
    """The oracle (CPU port, 1 thread) on a bounded sample of the same workload: the PU jobs of `n_ctus`
    interior CTUs (full +-128 windows) through hop_o_me_pu + hop_o_pred_inter.  Also cross-checks the GPU results."""
    from hoputil import oracle, p16, I16P
    O = oracle()
    bufs = [np.ascontiguousarray(ctx.ssref_download(c)) for c in range(3)]
    sy, sc = bufs[0].shape[1], bufs[1].shape[1]
    y00 = ctypes.cast(bufs[0].ctypes.data + (80 * sy + 80) * 2, I16P)
    cb00 = ctypes.cast(bufs[1].ctypes.data + (40 * sc + 40) * 2, I16P)
    cr00 = ctypes.cast(bufs[2].ctypes.data + (40 * sc + 40) * 2, I16P)
    Yh = Y.cpu().numpy()
    hctu = (H + 63) // 64
    r0 = min(hctu - 1, 4)
    sample = [r0 * wctu + min(wctu - 1, 5) + i for i in range(n_ctus)]
    sel = np.nonzero(np.isin(ctu_of_job, sample))[0]
    t0 = time.perf_counter()
    mism = 0
    for k in sel:
        j = jobs[k]
        org = np.ascontiguousarray(Yh[j["pu_y"]:j["pu_y"] + j["h"], j["pu_x"]:j["pu_x"] + j["w"]])
        out = (ctypes.c_int64 * 32)()
        O.hop_o_me_pu(p16(org), int(j["w"]), y00, sy, int(j["pu_x"]), int(j["pu_y"]), int(j["w"]), int(j["h"]), int(j["rng_left"]), int(j["rng_right"]),
                      int(j["rng_top"]), int(j["rng_bottom"]), int(j["off_x"]), int(j["off_y"]), int(j["pred_x"]), int(j["pred_y"]), int(j["n_amvp"]),
                      (ctypes.c_int * 4)(*[int(v) for v in j["amvp"]]), lc, 1, 1, 8, 3, out)
        r = res_gpu[k]
        if not out[3]:
            got = [int(r["mv_int"][0]), int(r["mv_int"][1]), int(r["sad"]), int(r["gt_flag"])] + [int(v) for v in r["gt"]] + [int(r["cost"])]
            want = [out[0], out[1], out[2], out[9]] + list(out)[10:18] + [out[18]]
            mism += got != want
            if is2n[k]:
                w, h = int(j["w"]), int(j["h"])
                a = [np.zeros((h, w), np.int16), np.zeros((h // 2, w // 2), np.int16), np.zeros((h // 2, w // 2), np.int16)]
                mvx, mvy = (out[19] << 2) + (out[21] << 1) + out[23], (out[20] << 2) + (out[22] << 1) + out[24]
                O.hop_o_pred_inter(y00, sy, cb00, cr00, sc, int(j["pu_x"]), int(j["pu_y"]), w, h, int(mvx), int(mvy), 1,
                                   (ctypes.c_int * 8)(*[int(v) for v in list(out)[10:18]]), 8, 8, p16(a[0]), p16(a[1]), p16(a[2]))
        else:
            mism += int(r["not_valid"]) != 1
    dt = time.perf_counter() - t0
    return {"value": len(sample) / dt, "unit": "CTU/s", "cores": 1, "kind": "port",
            "sample": "%d interior CTUs (row %d) of the same frame, %d PU jobs: oracle hop_o_me_pu (+ hop_o_pred_inter on 2Nx2N) single thread, %.1f s" % (len(sample), r0, len(sel), dt),
            "gpu_vs_oracle_mismatches": int(mism)}


def band_of_rank(frame_h, rows, rank):
    """(kept for tests/test_multi_rank.py) the frame cut into bands of `rows` CTU rows; rank k codes band k (mod the number of bands) as its picture.
    Returns (band index, first luma row, band height)."""
    hb = rows * 64
    nb = max(1, frame_h // hb)
    b = rank % nb
    return b, b * hb, hb


def tiles_of_rank(frame_w, frame_h, tile_w, tile_h, pictures, rank):
    """The independent pictures of a rank: every frame of the sequence is cut into tile_w x tile_h pictures (whole ones only, raster order) and the pictures are numbered
    through the frames; rank k codes pictures k * pictures ... (k + 1) * pictures - 1.  Returns [(frame, x0, y0), ...]."""
    tx, ty = max(1, frame_w // tile_w), max(1, frame_h // tile_h)
    out = []
    for i in range(pictures):
        t = rank * pictures + i
        out.append((t // (tx * ty), (t % (tx * ty)) % tx * tile_w, (t % (tx * ty)) // tx * tile_h))
    return out


def views_of_rank(n_views, world, rank):
    """BASELINE config 5: the 13 x 13 sub-aperture views of the light field, dealt round-robin to the ranks (view v -> rank v % world); every view is an independent ISS picture"""
    return [v for v in range(n_views) if v % world == rank]


def wavefront_ramp_ctus(cols, rows, lag):
    """CTUs a lag-`lag` wavefront of a cols x rows picture has retired when the number of rows in flight first reaches its maximum, and that maximum"""
    rif = min(rows, (cols + lag - 1) // lag)
    s_full = lag * (rif - 1)                                               # the wavefront step at which row rif - 1 starts
    done = sum(min(cols, max(0, s_full - lag * r)) for r in range(rows))
    return done, rif


class EncodeRun:
    """hop_encode_frame on a thread of its own (ctypes releases the GIL); the caller watches hop_encode_progress and ends the run with hop_encode_cancel"""
    def __init__(self, ctx, lag, mi):
        import threading
        self.ctx, self.out, self.err = ctx, None, None
        self.t0 = time.perf_counter()
        self.th = threading.Thread(target=self._run, args=(lag, mi), daemon=True)
        self.th.start()

    def _run(self, lag, mi):
        try:
            self.out = self.ctx.encode_frame(QP, mi, 0, None, wpp=1, wavefront_lag=lag)
        except BaseException as e:                                         # reported by the watcher
            self.err = e

    def wait_progress(self, target, deadline, poll=0.002):
        """until `target` CTUs are retired (-> (time, retired)) or the deadline passes / the run ends (-> (time, retired) with retired < target)"""
        while True:
            n = self.ctx.encode_progress()
            t = time.perf_counter()
            if n >= target or t >= deadline or not self.th.is_alive():
                return t, n
            time.sleep(poll)

    def finish(self):
        self.ctx.encode_cancel()
        self.th.join()
        if self.err is not None:
            raise self.err
        return self.out


def frame_planes(fw, fh, seed):
    """bench.py's frame: tests/hoputil.py:lenslet in numpy float64 on the host -- the SAME function oracle/make_golden24.py feeds the reference encoder with, so that
    the per-CTU RD costs of a run can be compared with the reference's cost.csv (the GPU generator of --kernels mode is not reproducible on a CPU)"""
    from hoputil import lenslet
    return lenslet(fw, fh, PITCH, seed)


def golden_costs(fh, Y, Cb, Cr):
    """the reference encoder's per-CTU RD costs (cost.csv) for this frame, if a golden made from the same planes is committed (tests/golden/encoder_frame_mi15_rows*.npz):
    (costs, meta, n_comparable) or (None, why, 0).  A golden of exactly this picture pins every CTU; one of a shorter full-width band of the same frame pins CTU row 0 only
    (the first row is coded alike in both: nothing of row 0 depends on where the picture ends further down)."""
    import glob, hashlib
    md5 = lambda a: hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()
    cols, cands = (Y.shape[1] + 63) // 64, []
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "encoder_frame_mi15_rows*.npz"))):
        g = np.load(path)
        meta = json.loads(g["meta"].tobytes().decode())
        if meta["W"] != Y.shape[1] or meta["H"] > fh: continue
        cands.append((meta["H"] == fh, meta["H"], g["cost"], meta, path))
    if not cands: return None, "no golden for this frame geometry", 0
    exact, H, cost, meta, path = max(cands, key=lambda c: (c[0], c[1]))
    if md5(Y[:H]) != meta["y_md5"] or md5(Cb[:H // 2]) != meta["cb_md5"] or md5(Cr[:H // 2]) != meta["cr_md5"]:
        return None, "the frame generated on this host differs from the golden's input (numpy's sin / cos are not bit-reproducible across CPUs): comparison skipped", 0
    meta["path"] = os.path.relpath(path, ROOT)
    return cost, meta, (len(cost) if exact else cols)


def cpu_baseline_reference(Y, Cb, Cr, crop_w, crop_h, budget_s):
    """The unmodified reference encoder (oracle/_ref/TAppEncoderRef, built in the build container from /root/reference by oracle/Makefile.ref; it travels with the repository
    snapshot) on one host core: an interior crop of the same frame, HOP configuration, --MIsize=15, QP 32.  CTU/s = CTUs / its own wall time.  Falls back to the port."""
    import subprocess, tempfile
    from hoputil import hop_encoder_args
    exe = os.path.join(ROOT, "oracle", "_ref", "TAppEncoderRef")
    if not os.path.exists(exe): return None
    y0, x0 = (Y.shape[0] // 2) // 64 * 64, (Y.shape[1] // 2) // 64 * 64
    cy, cb, cr = Y[y0:y0 + crop_h, x0:x0 + crop_w], Cb[y0 // 2:(y0 + crop_h) // 2, x0 // 2:(x0 + crop_w) // 2], Cr[y0 // 2:(y0 + crop_h) // 2, x0 // 2:(x0 + crop_w) // 2]
    n = ((crop_w + 63) // 64) * ((crop_h + 63) // 64)
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "in.yuv"), "wb").write(cy.astype(np.uint8).tobytes() + cb.astype(np.uint8).tobytes() + cr.astype(np.uint8).tobytes())
        t0 = time.perf_counter()
        try:
            r = subprocess.run([exe] + hop_encoder_args(crop_w, crop_h, mi=PITCH), cwd=td, capture_output=True, text=True, timeout=budget_s)
        except subprocess.TimeoutExpired:
            return {"error": "the reference encoder did not finish %d CTUs in %.0f s" % (n, budget_s)}
        dt = time.perf_counter() - t0
        if r.returncode != 0: return {"error": "TAppEncoderRef exited with %d" % r.returncode}
        tot = [ln.strip() for ln in r.stdout.split("\n") if "Total Time" in ln]
    return {"value": n / dt, "unit": "CTU/s", "cores": 1, "kind": "reference", "host_cpus": os.cpu_count(),
            "sample": "oracle/_ref/TAppEncoderRef (the unmodified reference encoder, -O3, single-threaded by construction) on a %dx%d crop (%d CTUs) from the middle of the same frame as one "
                      "picture, cfg/3DHencoder_intra_main.cfg semantics, --MIsize=%d, QP %d: %.1f s wall (%s); the crop's border CTUs see truncated SS windows, so this OVERstates "
                      "the steady-state rate of interior CTUs of the 7728-wide frame" % (crop_w, crop_h, n, PITCH, QP, dt, tot[0] if tot else "")}


def cpu_baseline_port(Y, Cb, Cr, n_ctus):
    """fallback when the reference binary did not travel: the same RD spine over the CPU restatement (oracle/libhop_spine_cpu.so, built by __graft_entry__.build()), one core"""
    path = os.path.join(ROOT, "oracle", "libhop_spine_cpu.so")
    if not os.path.exists(path): return {"error": "neither oracle/_ref/TAppEncoderRef nor oracle/libhop_spine_cpu.so is present (run __graft_entry__.build())"}
    L = ctypes.CDLL(path)
    L.hop_spine_cpu_encode.restype = ctypes.c_long
    L.hop_spine_cpu_encode.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3 + [ctypes.c_char_p] + [ctypes.c_void_p] * 8
    W, H = 1024, 128
    a = [np.ascontiguousarray(p, np.int16) for p in (Y[:H, :W], Cb[:H // 2, :W // 2], Cr[:H // 2, :W // 2])]
    cost = np.zeros(((W + 63) // 64) * ((H + 63) // 64), np.float64)
    t0 = time.perf_counter()
    L.hop_spine_cpu_encode(W, H, QP, PITCH, n_ctus, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, None, cost.ctypes.data, None, None, None, None, None, None, None)
    dt = time.perf_counter() - t0
    return {"value": n_ctus / dt, "unit": "CTU/s", "cores": 1, "kind": "port", "host_cpus": os.cpu_count(),
            "sample": "the first %d CTUs of the frame's top-left %dx%d through the same RD spine over the CPU restatement, one thread, %.1f s (top-row CTUs: cheaper than interior ones)" % (n_ctus, W, H, dt)}


PROF_NAMES = {0: "k_ss_search / k_ss_family", 1: "k_frac", 2: "k_gt_search", 3: "k_pred_inter", 4: "k_ssref_commit", 5: "k_distortion",
              6: "transform-unit leaf step + candidate walks (transform, estBit, RDOQ, counted bits, inverse, decisions)", 7: "k_intra (rough search + predictors)", 8: "k_rdoq (staged form)",
              9: "k_coeff_bits + CU-level counting",
              12: "k_inter_walk 8x8 (SS/GT candidate: residual quadtree, finish, CU bits)", 13: "k_inter_walk 16x16", 14: "k_inter_walk 32x32", 15: "k_inter_walk 64x64",
              16: "k_intra_walk 8x8 2Nx2N (intra candidate: luma search, chroma search, CU bits)", 17: "k_intra_walk 16x16", 18: "k_intra_walk 32x32", 19: "k_intra_walk 64x64", 20: "k_intra_walk 8x8 NxN"}


def profiled_pass(hp, local, Y, Cb, Cr, w, h, slots, lag):
    """the roofline object's numbers: a separate, untimed encode of the frame's top-left w x h as one picture with HIP events around every launch (hop_profile_*)"""
    pctx = hp.Context(w, h, device=local, slots=slots)
    pctx.upload_orig(np.ascontiguousarray(Y[:h, :w]), np.ascontiguousarray(Cb[:h // 2, :w // 2]), np.ascontiguousarray(Cr[:h // 2, :w // 2]))
    L = pctx.L
    L.hop_profile_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.hop_profile_reset.argtypes = [ctypes.c_void_p]
    L.hop_profile_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    pctx._chk(L.hop_profile_reset(pctx.h), "profile_reset"); pctx._chk(L.hop_profile_enable(pctx.h, 1), "profile_enable")
    t0 = time.perf_counter()
    pctx.encode_frame(QP, PITCH, 0, None, wpp=1, wavefront_lag=lag)
    dt = time.perf_counter() - t0
    prof = {}
    for kid, name in PROF_NAMES.items():
        la, ms, un = ctypes.c_uint64(), ctypes.c_double(), ctypes.c_uint64()
        pctx._chk(L.hop_profile_read(pctx.h, kid, ctypes.byref(la), ctypes.byref(ms), ctypes.byref(un)), "profile_read")
        prof[name] = {"launches": la.value, "total_ms": ms.value, "units": un.value}
    pctx._chk(L.hop_profile_enable(pctx.h, 0), "profile_disable")
    pctx.close()
    return prof, ((w + 63) // 64) * ((h + 63) // 64), dt


def view_planes(W, H, u, v, seed=7):
    """BASELINE config 5: sub-aperture view (u, v) of a synthetic 13 x 13 light field -- one scene texture seen with a disparity of (u - 6, v - 6) x 0.8 samples, noise of its
    own; 8 bit 4:2:0 (numpy, host)"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    sx, sy = xx + (u - 6) * 0.8, yy + (v - 6) * 0.8

    def tex(ax, ay):
        t = np.zeros_like(ax)
        for _ in range(5):
            f = rng.uniform(0.01, 0.25, 2); ph = rng.uniform(0, 2 * np.pi, 2)
            t += rng.uniform(0.4, 1.0) * np.sin(ax * f[0] + ph[0]) * np.cos(ay * f[1] + ph[1])
        return t / 5.0
    ty = tex(sx, sy); tb = tex(sx[::2, ::2], sy[::2, ::2]); tr = tex(sx[::2, ::2], sy[::2, ::2])
    noise = np.random.default_rng(1000 + 13 * v + u).normal(0, 2.0, (H, W))
    Y = np.clip(np.rint(128 + 100 * ty + noise), 0, 255).astype(np.int16)
    return Y, np.clip(np.rint(128 + 60 * tb), 0, 255).astype(np.int16), np.clip(np.rint(128 + 60 * tr), 0, 255).astype(np.int16)


def views_pass(hp, local, rank, world, slots, lag, budget_s, n_views=169, W=624, H=432):
    """BASELINE config 5 (cfg/3DHencoder_intra_main.cfg on the 13 x 13 sub-aperture views, every frame an ISS picture of its own): this rank's views (round-robin) side by side in
    one stacked context, coded by ONE hop_encode_frame -- the rows of all pictures form one wavefront, a batch serves a CTU of every row in flight.  Runs until the stack is
    coded or the budget is spent; the rate is CTUs retired / wall time from the start of the encode (the wavefront's ramp and drain included when it finishes)."""
    mine = views_of_rank(n_views, world, rank)
    pics = [view_planes(W, H, v % 13, v // 13) for v in mine]
    ctx = hp.Context(W, H, device=local, pictures=len(pics), slots=slots)
    ctx.upload_orig(ctx.stack([p[0] for p in pics]), ctx.stack([p[1] for p in pics], True), ctx.stack([p[2] for p in pics], True))
    ctx.sync()
    n_ctu = ((W + 63) // 64) * ((H + 63) // 64) * len(pics)
    run = EncodeRun(ctx, lag, 16)
    t, n = run.wait_progress(n_ctu, time.perf_counter() + budget_s, poll=0.01)
    dt = t - run.t0
    run.finish()
    st = ctx.encode_stats()
    ctx.close()
    return {"workload": "BASELINE config 5: %d of the %d sub-aperture views (%dx%d, synthetic light field) per GPU as independent ISS pictures in one stacked context, "
                        "cfg/3DHencoder_intra_main.cfg semantics, QP %d" % (len(pics), n_views, W, H, QP),
            "pictures_per_gpu": len(pics), "ctus": n_ctu, "ctus_retired": int(n), "seconds": dt, "ctu_per_s_per_gpu": n / dt if dt > 0 else 0.0, "finished": bool(n >= n_ctu),
            "rendezvous_rounds": st["rendezvous"]["rounds"]}


def encode_main(args):
    """The default: BASELINE.json's metric on BASELINE's configuration -- the 7728 x 5368 lenslet frame coded as ONE picture, exactly as TEncSlice::compressSlice codes it with
    cfg/3DHencoder_intra_main.cfg --MIsize=15 and WaveFrontSynchro (one substream per CTU row): every candidate of TEncCu::xCompressCU for every CTU, the SS reference starting at
    the sentinel and growing CU by CU, coder contexts carried from CU to CU, the CTU rows as a wavefront of lag --lag.  The picture takes minutes, so it is coded CONTINUOUSLY (one
    hop_encode_frame on a thread of its own) and a STEP is a fixed quantum of retired CTUs (--step-ctus, default one CTU row = 121) read from hop_encode_progress: untimed are
    the ramp (until the wavefront holds its maximum of rows) and `--warmup` steps; then `--steps` steps are timed; then the run is cancelled.  A wall-clock budget (--budget-s,
    measured from process start) ends the timed region early if need be: `steps` in the JSON is what was timed.  With --gpus N every rank codes its own frame of the sequence."""
    t_proc = time.perf_counter() - (time.time() - PROC_T0)
    import threading
    timp = threading.Thread(target=lambda: __import__("torch"), daemon=True); timp.start()     # (the first import on a fresh box takes a minute or two: beside the frame generation)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    deadline = t_proc + args.budget_s
    fw, fh = args.width, min(args.height, args.rows * 64)
    shard = args.shard_rows and world > 1                                    # ONE picture over all ranks (its CTU rows dealt round-robin) instead of a picture per rank
    Y, Cb, Cr = frame_planes(fw, args.height, 2 + (0 if shard else rank))    # frame `rank` of the synthetic sequence: the same lenslet geometry, another texture seed
    Y, Cb, Cr = np.ascontiguousarray(Y[:fh]), np.ascontiguousarray(Cb[:fh // 2]), np.ascontiguousarray(Cr[:fh // 2])
    timp.join()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libhophip has no CPU path")
    ndev = torch.cuda.device_count()
    shared = world > ndev
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo" if shared else "nccl", **({} if shared else {"device_id": dev}))
    hp = _hophip()
    cols, rows = (fw + 63) // 64, (fh + 63) // 64
    n_ctu = cols * rows
    lag = min(args.lag, cols)
    ramp_ctus, rif = wavefront_ramp_ctus(cols, rows, lag)
    Q = args.step_ctus if args.step_ctus > 0 else cols
    if shard:                                                                # hop_encode_progress counts the CTUs THIS rank retired: its share of a step, of the ramp
        Q = max(1, Q // world); ramp_ctus = max(1, ramp_ctus // world)

    def barrier():
        torch.cuda.synchronize()
        if world > 1: dist.barrier()

    # ---- rank 0, before the timed run and bounded: the CPU baseline and the profiled pass behind the roofline object ----
    extras = {}
    if rank == 0:
        try:
            cb = cpu_baseline_reference(Y, Cb, Cr, args.cpu_crop[0], args.cpu_crop[1], args.cpu_budget_s) if not args.no_cpu else {"skipped": True}
            if cb is None: cb = cpu_baseline_port(Y, Cb, Cr, 6)
        except Exception as e:
            cb = {"error": repr(e)}
        extras["cpu_baseline"] = cb
        try:
            extras["prof"] = profiled_pass(hp, local, Y, Cb, Cr, min(fw, args.profile_w), min(fh, args.profile_h), args.slots, lag)
        except Exception as e:
            extras["prof_error"] = repr(e)
    ctx = hp.Context(fw, fh, device=local, slots=args.slots)
    ctx.upload_orig(Y, Cb, Cr)
    ctx.sync()
    gather = None
    if shard:
        spec = importlib.util.spec_from_file_location("hop_shard", os.path.join(ROOT, "hevc-hop_amd", "shard.py"))
        shm = importlib.util.module_from_spec(spec); spec.loader.exec_module(shm)
        # RCCL between the GPUs of the node (gloo when ranks share a GPU in a rehearsal), on a process group of its own: the exchange runs beside this thread's barriers
        gather = shm.TorchAllgather(None if shared else dev, group=dist.new_group(list(range(world))))
        ctx.set_shard(rank, world, gather)
    barrier()
    t_setup = time.perf_counter() - t_proc
    # ---- the continuously running picture ----
    run = EncodeRun(ctx, lag, PITCH)
    ramp_deadline = min(deadline, time.perf_counter() + args.ramp_frac * max(1.0, deadline - time.perf_counter()))
    t_r, n_r = run.wait_progress(min(ramp_ctus, n_ctu), ramp_deadline)          # ramp: the wavefront fills
    t_w, n_w = run.wait_progress(min(n_r + args.warmup * Q, n_ctu), deadline)   # warm-up steps
    if world > 1: dist.barrier()                                                # all ranks enter the timed region together (their pictures keep running meanwhile)
    n0 = ctx.encode_progress(); t0 = time.perf_counter()
    marks = [(t0, n0)]
    for k in range(args.steps):
        t, n = run.wait_progress(min(n0 + (k + 1) * Q, n_ctu), deadline)
        if n < n0 + (k + 1) * Q: break                                          # budget spent (or the picture ended): the steps so far count
        marks.append((t, n))
    steps_done = len(marks) - 1
    if world > 1:
        sd = torch.tensor([steps_done], dtype=torch.int64, device="cpu" if shared else dev)
        dist.all_reduce(sd, op=dist.ReduceOp.MIN)
        steps_done = int(sd.item())
    if steps_done < 1:
        t, n = run.wait_progress(n_ctu, time.perf_counter())                    # nothing completed inside the budget: report what there is, as a fraction of a step
        marks.append((t, n)); steps_done = 0
    t1, n1 = marks[steps_done] if steps_done >= 1 else marks[-1]
    dt, ctus_timed = t1 - t0, n1 - n0
    cost, bits, dist_, parts, ncand = run.finish()
    ctx.sync(); barrier()
    stats = ctx.encode_stats()
    retired = int(ctx.encode_progress())
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        cc = torch.tensor([ctus_timed], dtype=torch.int64, device="cpu" if shared else dev)
        dist.all_reduce(cc, op=dist.ReduceOp.SUM)
        ctus_all = int(cc.item())
    else:
        ctus_all = ctus_timed
    if rank == 0:
        value = ctus_all / dt if dt > 0 else 0.0
        # ---- parity of what was measured: the retired CTUs' RD costs against the reference encoder's cost.csv for the same frame ----
        gold, meta, m = golden_costs(fh, Y, Cb, Cr)
        done = cost > 0
        if gold is not None:
            sel = done[:m]
            parity = {"reference": "cost.csv of the unmodified reference encoder for %s this picture, %s (oracle/make_golden24.py)" % ("exactly" if m == len(cost) else "the first CTU row of", meta["path"]),
                      "ctus_compared": int(sel.sum()), "mismatches": int(np.sum(cost[:m][sel] != gold[:m][sel])), "mismatch_ctus": [int(a) for a in np.nonzero((cost[:m] != gold[:m]) & sel)[0][:8]],
                      "mismatch_costs_here_reference": [[int(a), float(cost[a]), float(gold[a])] for a in np.nonzero((cost[:m] != gold[:m]) & sel)[0][:8]], "reference_one_core_ctu_per_s": meta.get("ctu_per_s_one_core"),
                      "reference_cpu": meta.get("cpu")}
        else:
            parity = {"reference": None, "note": meta}
        prof, dom, roof = extras.get("prof"), None, None
        if prof:
            pk, prof_ctus, prof_s = prof
            # the dominant KERNEL: the classes of one kernel function (the CU sizes of a walk kernel) count together, as rocprofv3's per-kernel statistics do
            byfn = {}
            for name, v in pk.items():
                f = byfn.setdefault(name.split(" ")[0], {"launches": 0, "total_ms": 0.0})
                f["launches"] += v["launches"]; f["total_ms"] += v["total_ms"]
            dom = max(byfn, key=lambda k: byfn[k]["total_ms"])
            p = byfn[dom]
            avg_ms = p["total_ms"] / max(1, p["launches"])
            ctus_per_launch = prof_ctus / max(1, p["launches"])
            achieved = ALGO_BYTES_PER_CTU * ctus_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms else 0.0
            traffic, tsrc = None, None
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "r03_traffic.json")))
                traffic, tsrc = tj["kernels"][dom]["hbm_bytes_per_launch"], tj["source"]
            except Exception:
                pass
            roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                    "avg_launch_ms": avg_ms, "ctus_per_launch": ctus_per_launch, "launches_per_ctu": sum(v["launches"] for v in pk.values()) / prof_ctus,
                    "algorithmic_bytes_per_ctu": ALGO_BYTES_PER_CTU,
                    "note": "from a separate profiled pass over the frame's top-left %dx%d as one picture (%d CTUs, %.1f s, HIP events around every launch, one stream): the encode is bound by the LENGTH of "
                            "its dependent chains (serial coder walks inside the candidate evaluations: SQ_WAIT_ANY 85 %% of the walk kernels' wave cycles, profiles/r03_encode_pmc.json), not by HBM or VALU "
                            "throughput -- see rendezvous / request_ms; per-kernel times of the timed command itself: profiles/r03_bench_kernel_stats.csv" % (min(fw, args.profile_w), min(fh, args.profile_h), prof_ctus, prof_s)}
        rv = stats.get("rendezvous", {"rounds": 0, "requests": 0})
        out = {
            "metric": "CTUs/sec all-intra 7728x5368 lenslet @QP32 (RD search of TEncSlice::compressSlice; per-CTU RD costs checked against the reference encoder's cost.csv in `parity`)",
            "value": value, "unit": "CTU/s", "n_gpus": world, "steps": steps_done, "warmup": args.warmup, "ms_per_step": (dt / steps_done * 1e3) if steps_done else None,
            "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None, "dtype": "i16+f64", "data": "synthetic",
            "config": {"workload": "synthetic lenslet %dx%d (pitch %d, tests/hoputil.py:lenslet seed 2 + rank), QP%d, coded as ONE picture per GPU with cfg/3DHencoder_intra_main.cfg semantics and --MIsize=%d "
                                   "(ISS slice, SS +-128 full search with FEN, half/quarter-pel, GT search, merge / AMVP / micro-image candidates, AMP, intra 35 modes with RQT, RDOQ, transform skip, "
                                   "CABAC-counted bits), WaveFrontSynchro with one substream per CTU row, rows as a lag-%d wavefront (at most %d rows in flight); the picture is coded continuously, "
                                   "a step = %d retired CTUs" % (fw, fh, PITCH, QP, PITCH, lag, rif, Q),
                       "picture": [fw, fh], "ctus_per_picture": n_ctu, "ctus_per_step": Q, "candidate_slots": args.slots, "rows_in_flight_max": rif,
                       "parallelism": ("ONE picture over %d GPUs: CTU rows dealt round-robin, every wavefront step's finished CTUs exchanged by an all-gather (%s; %d exchanges, %.1f MB "
                                       "received per rank); ctus_per_step is a rank's share" % (world, "gloo, ranks share a GPU" if shared else "RCCL", gather.calls, gather.bytes / 1e6)) if shard else
                                      "one picture per GPU x%d GPUs (frames of the sequence; no data-path collective)" % world},
            "timed_region": {"ctus": ctus_all, "seconds": dt, "ctus_retired_before": n0, "ramp_ctus": n_r, "ramp_s": t_r - run.t0, "warmup_ctus": n_w - n_r, "warmup_s": t_w - t_r,
                             "retired_total": retired, "budget_s": args.budget_s, "setup_s": t_setup,
                             "note": "steps requested %d; steps timed %d (a wall-clock budget measured from process start ends the timed region early)" % (args.steps, steps_done)},
            "parity": parity,
            "roofline": roof,
            "kernels": extras["prof"][0] if prof else {"error": extras.get("prof_error")},
            "request_ms": {k: v for k, v in stats.items() if k != "rendezvous"},
            "request_note": "host wall time per kind of request of the whole run (ramp included), summed over the batches (one batch serves all CTUs in flight)",
            "rendezvous": {"rounds": rv["rounds"], "requests": rv["requests"], "avg_batch": rv["requests"] / max(1, rv["rounds"]), "serve_ms": rv.get("serve_ms", 0.0), "run_ms": rv.get("run_ms", 0.0),
                           "rounds_per_retired_ctu": rv["rounds"] / max(1, retired)},
            "wavefront_visibility": {"searches_reaching_below": rv.get("searches_reaching_below"), "first_ctu_reaching_below": rv.get("first_ctu_reaching_below"),
                                     "searches_reaching_above": rv.get("searches_reaching_above"), "first_ctu_reaching_above": rv.get("first_ctu_reaching_above"),
                                     "note": "motion searches whose window covered samples the wavefront and the reference's raster order see differently (committed samples of "
                                             "the CTU rows below / not yet coded samples of the rows above): where the run's costs can leave the reference's (DESIGN.md section 5)"},
            "candidates": ncand, "cost_sum_retired": float(cost[done].sum()),
            "cpu_baseline": extras.get("cpu_baseline"),
        }
    ctx.close()
    if gather is not None: gather.close()
    if rank == 0:
        try: open(os.path.join(ROOT, "gpurun_out", "bench_last.json"), "w").write(json.dumps(out))      # (kept in case the optional pass below is cut short)
        except Exception: pass
    # ---- the second workload (BASELINE config 5), when the wall-clock budget still has room for it: all ranks, each with its share of the views ----
    t_left = t_proc + args.total_s - time.perf_counter()
    want_views = args.views > 0 and t_left > args.views_min_s and not shard
    if world > 1:
        wv = torch.tensor([1 if want_views else 0], dtype=torch.int64, device="cpu" if shared else dev)
        dist.all_reduce(wv, op=dist.ReduceOp.MIN)
        want_views = bool(wv.item())
    if want_views:
        try:
            vp = views_pass(hp, local, rank, world, args.slots, lag, min(args.views_budget_s, t_left - 40.0), n_views=args.views)
        except Exception as e:
            vp = {"error": repr(e), "ctu_per_s_per_gpu": 0.0}
        if world > 1:
            vr = torch.tensor([vp.get("ctu_per_s_per_gpu", 0.0)], dtype=torch.float64, device="cpu" if shared else dev)
            dist.all_reduce(vr, op=dist.ReduceOp.SUM)
            vp["ctu_per_s_all_gpus"] = float(vr.item())
        if rank == 0: out["cfg5_views"] = vp
    elif rank == 0:
        out["cfg5_views"] = {"skipped": "no room left in the wall-clock budget (%.0f s left); `bench.py --views 169 --steps 1 --warmup 0` measures it alone" % t_left}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def spawn_ranks(argv, n):
    """`bench.py --gpus N` without a launcher: N rank processes through torch.distributed.run, started BEFORE this process touches torch or HIP; rank 0's JSON line is
    relayed.  (Never re-exec after GPU initialisation: this process stays a plain parent.)"""
    import socket, subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.split("\n") if ln.startswith("{") and '"metric"' in ln]
    if lines: print(lines[-1], flush=True)
    sys.exit(r.returncode if not lines else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=FRAME_W)     # smaller pictures only for rehearsal; the JSON names them
    ap.add_argument("--height", type=int, default=FRAME_H)
    ap.add_argument("--rows", type=int, default=84, help="CTU rows of the picture: 84 = the whole 7728x5368 frame; fewer = its full-width top band as one picture")
    ap.add_argument("--step-ctus", type=int, default=0, help="retired CTUs that make one step (0: one CTU row of the picture)")
    ap.add_argument("--budget-s", type=float, default=430.0, help="wall-clock budget from process start: the timed region ends there at the latest (the driver's limit is 600 s)")
    ap.add_argument("--ramp-frac", type=float, default=0.55, help="at most this share of the time left when the picture starts goes into the ramp")
    ap.add_argument("--slots", type=int, default=48, help="candidate slots per context (the SS/GT candidates of a CU side by side); 0 = one after the other")
    ap.add_argument("--lag", type=int, default=8, help="wavefront lag in CTUs: row r codes CTU c once row r - 1 has finished CTU c + lag - 1.  8: identical to the reference on the whole frame (10 157 CTUs compared, --steps 76); 5 is faster (76 against 53 CTU/s) and leaves the reference at CTU 3079 (DESIGN.md section 5)")
    ap.add_argument("--shard-rows", action="store_true", help="--gpus N > 1: ONE picture over all ranks, its CTU rows dealt round-robin with a hand-off after every wavefront step "
                                                               "(hop_encode_set_shard; strong scaling) instead of one picture per rank")
    ap.add_argument("--profile-w", type=int, default=1024); ap.add_argument("--profile-h", type=int, default=64)
    ap.add_argument("--cpu-crop", type=int, nargs=2, default=[256, 192], help="crop (from the middle of the frame) the reference CPU encoder is timed on")
    ap.add_argument("--cpu-budget-s", type=float, default=60.0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg (development runs)")
    ap.add_argument("--views", type=int, default=169, help="BASELINE config 5 as a second, separately named figure (cfg5_views): this many 624x432 sub-aperture views as independent pictures; 0 = off")
    ap.add_argument("--views-budget-s", type=float, default=90.0); ap.add_argument("--views-min-s", type=float, default=130.0, help="room the wall-clock budget must still have for the views pass")
    ap.add_argument("--total-s", type=float, default=560.0, help="everything is printed within this many seconds from process start (the driver's limit is 600 s)")
    ap.add_argument("--cpu-ctus", type=int, default=None, help="--kernels: CTUs of the bounded cpu_baseline sample")
    ap.add_argument("--kernels", action="store_true", help="kernel-throughput mode of round 1: the search kernels over a frozen, fully reconstructed SS reference (not the encode metric)")
    ap.add_argument("--rqt", action="store_true", help="--kernels: run the whole residual-quadtree search of every 2Nx2N CU instead of its leaf step")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(sys.argv[1:], args.gpus)
    if args.kernels:
        if args.cpu_ctus is None: args.cpu_ctus = 10
        if args.steps == 4 and args.warmup == 1: args.steps, args.warmup = 3, 1
        kernels_main(args)
    else:
        encode_main(args)


if __name__ == "__main__":
    main()
