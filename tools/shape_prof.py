#!/usr/bin/env python3
"""tools/shape_prof.py -- per-PU-shape timing of the hot-path kernels on a small frame (developer tool).
Usage (GPU box): python tools/shape_prof.py [stage]"""
import ctypes, importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import torch
from bench import lenslet_torch, cu_rects, _hophip
from hoputil import lambda_for_qp
hp = _hophip()
W, H = 2048, 1024
dev = torch.device("cuda", 0)
Y, Cb, Cr = lenslet_torch(W, H, 15, 2, dev)
ctx = hp.Context(W, H); L = ctx.L
ctx.upload_orig(Y.cpu().numpy(), Cb.cpu().numpy(), Cr.cpu().numpy())
rects = cu_rects(W, H); d_rects = torch.from_numpy(rects).to(dev)
ctx._chk(L.hop_ssref_commit_cus_device(ctx.h, len(rects), d_rects.data_ptr(), Y.data_ptr(), Cb.data_ptr(), Cr.data_ptr()), "commit")
lam, lc = lambda_for_qp(32)
wctu, hctu = W // 64, H // 64
pred = (ctypes.c_int * 2)(0, -60); amvp = (ctypes.c_int * 4)(0, -60, -60, 0)
cap = 1275 * wctu * hctu
jobs = np.zeros(cap, hp.PU_JOB_DTYPE); n = 0
for a in range(wctu * hctu):
    n += L.hop_enumerate_ctu_jobs(W, H, a, 128, pred, 2, amvp, lc, 3, 1, jobs.ctypes.data + n * jobs.itemsize, None, cap - n)
jobs = jobs[:n]
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 3
shapes = sorted(set(zip(jobs["w"].tolist(), jobs["h"].tolist())), key=lambda s: (-s[0] * s[1], -s[0]))
if len(sys.argv) > 2:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]]
print("shape      njobs   ss_ms  frac_ms   gt_ms | ss Tsad/s  gt Msamp/s")
for (w, h) in shapes:
    sel = np.ascontiguousarray(jobs[(jobs["w"] == w) & (jobs["h"] == h)])
    dj = torch.from_numpy(sel.view(np.uint8)).to(dev)
    dr = torch.zeros(len(sel) * hp.PU_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    ctx._chk(L.hop_me_search_device(ctx.h, len(sel), dj.data_ptr(), dr.data_ptr(), stage), "me"); ctx.sync()
    L.hop_profile_reset(ctx.h); L.hop_profile_enable(ctx.h, 1)
    ctx._chk(L.hop_me_search_device(ctx.h, len(sel), dj.data_ptr(), dr.data_ptr(), stage), "me")
    ms = []
    for k in range(3):
        la, t, un = ctypes.c_uint64(), ctypes.c_double(), ctypes.c_uint64()
        L.hop_profile_read(ctx.h, k, ctypes.byref(la), ctypes.byref(t), ctypes.byref(un)); ms.append(t.value)
    L.hop_profile_enable(ctx.h, 0)
    win = (sel["rng_right"] - sel["rng_left"] + 1).clip(0).astype(np.float64) * (sel["rng_bottom"] - sel["rng_top"] + 1).clip(0)
    sads = float(np.sum(win * w * h / (2 if h > 8 else 1) / 2))
    it = 0; mm = min(w, h)
    while mm > 1 and it < 6: it += 1; mm //= 2
    samples = len(sel) * 3.0 * it * 56 * w * h
    print("%2dx%-2d  %8d %7.2f %7.2f %8.2f | %8.2f  %9.1f" % (w, h, len(sel), ms[0], ms[1], ms[2], sads / ms[0] / 1e9, samples / max(ms[2], 1e-9) / 1e3))
