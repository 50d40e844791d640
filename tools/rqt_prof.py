#!/usr/bin/env python3
"""tools/rqt_prof.py -- throughput of hop_rqt (row a8b, the whole residual-quadtree search) per CU size on synthetic residuals
(developer tool, GPU box).  Host-array entry point: the time includes staging the jobs and fetching results / levels."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
from bench import _hophip
hp = _hophip()
W, H = (7680, 5376) if "--full" in sys.argv else (4096, 2048)     # --full: the benchmark frame size rounded to CTUs
rng = np.random.default_rng(3)
yy, xx = np.mgrid[0:H, 0:W]
tex = (40 * np.sin(xx * 0.21) * np.cos(yy * 0.17) + rng.normal(0, 6, (H, W))).astype(np.int16)
org = [(128 + tex).astype(np.int16), (128 + tex[::2, ::2] // 2).astype(np.int16), (128 - tex[::2, ::2] // 2).astype(np.int16)]
ctx = hp.Context(W, H)
ctx.upload_orig(*org)
for comp in range(3):
    ctx.plane_upload("pred", comp, np.full(org[comp].shape, 128, np.int16))
snap = np.zeros((1, hp.CABAC_CTX_BYTES), np.uint8)
import ctypes
ctx.L.hop_cabac_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
ctx.L.hop_cabac_init(snap.ctypes.data, 3, 32)
LAM = 0.57 * 2.0 ** ((32 - 12) / 3.0)
print("CU     CUs   host s  device s  kCU/s(dev)  Msamples/s(dev)")
keep = []
for lg in (3, 4, 5, 6):
    S = 1 << lg
    xs, ys = np.meshgrid(np.arange(0, W, S), np.arange(0, H, S))
    n = xs.size
    jobs = np.zeros(n, hp.RQT_JOB_DTYPE)
    jobs["x"], jobs["y"], jobs["log2_cu"], jobs["ctx_index"] = xs.ravel(), ys.ravel(), lg, 0
    jobs["qp_scaled"] = (32, 31, 31); jobs["sign_hide"] = 1; jobs["use_ts"] = 1; jobs["log2_max_tu"] = 5
    jobs["log2_min_tu_in_cu"] = {3: 2, 4: 2, 5: 3, 6: 4}[lg]
    jobs["lambda_rd"] = LAM; jobs["lambda_rdoq"] = (LAM, LAM / 1.26, LAM / 1.26); jobs["dist_weight"] = (1.26, 1.26)
    for it in range(2):
        t0 = time.perf_counter()
        res, co, cx = ctx.rqt(jobs, snap)
        dt = time.perf_counter() - t0
    # device-resident form: jobs, snapshot, results and levels stay in HBM
    dev = torch.device("cuda", 0)
    dj = torch.from_numpy(jobs.view(np.uint8)).to(dev); ds = torch.from_numpy(snap).to(dev)
    dr = torch.zeros(n * hp.RQT_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev); dc = torch.zeros(n * S * S * 3 // 2, dtype=torch.int32, device=dev)
    ctx.L.hop_rqt_device.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 6
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx._chk(ctx.L.hop_rqt_device(ctx.h, n, dj.data_ptr(), jobs[:1].ctypes.data, ds.data_ptr(), dr.data_ptr(), dc.data_ptr(), None), "rqt_device")
        ctx.sync(); dtd = time.perf_counter() - t0
    assert np.array_equal(np.frombuffer(dr.cpu().numpy().tobytes(), hp.RQT_RESULT_DTYPE)["bits"], res["bits"]) and np.array_equal(dc.cpu().numpy(), co)
    keep.append((n, dj, jobs[:1].copy(), dr, dc, res["bits"].copy()))
    print("%2dx%-2d %7d %6.3f %8.3f %10.1f %12.1f   (mean depth %.2f, cbf %.0f%%)" % (S, S, n, dt, dtd, n / dtd / 1e3, n * S * S * 1.5 / dtd / 1e6,
          float(np.mean([r["tr_idx"][:S * S // 16].mean() for r in res[:2000]])), 100.0 * float(np.mean(res["cbf"][:, 0, 0] != 0))))

# all four classes at once (hop_rqt_device_classes: one stream per class)
k = len(keep)
ns = (ctypes.c_int * k)(*[t[0] for t in keep])
pj = (ctypes.c_void_p * k)(*[t[1].data_ptr() for t in keep]); pr = (ctypes.c_void_p * k)(*[t[3].data_ptr() for t in keep]); pc = (ctypes.c_void_p * k)(*[t[4].data_ptr() for t in keep])
cls = np.concatenate([t[2] for t in keep])
for t in keep: t[3].zero_()
ctx.L.hop_rqt_device_classes.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 7
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx._chk(ctx.L.hop_rqt_device_classes(ctx.h, k, ns, pj, cls.ctypes.data, ds.data_ptr(), pr, pc, None), "rqt_device_classes")
    ctx.sync(); dta = time.perf_counter() - t0
for t in keep:
    assert np.array_equal(np.frombuffer(t[3].cpu().numpy().tobytes(), hp.RQT_RESULT_DTYPE)["bits"], t[5])
print("all four sizes of the frame concurrently: %.3f s" % dta)
