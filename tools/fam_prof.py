#!/usr/bin/env python3
"""tools/fam_prof.py -- SS search time per CU size, CU families on / off (developer tool, GPU box)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
from bench import lenslet_torch, cu_rects, _hophip
from hoputil import lambda_for_qp
hp = _hophip()
W, H = 2048, 1024
dev = torch.device("cuda", 0)
Y, Cb, Cr = lenslet_torch(W, H, 15, 2, dev)
lam, lc = lambda_for_qp(32)
wctu, hctu = W // 64, H // 64
pred = (ctypes.c_int * 2)(0, -60); amvp = (ctypes.c_int * 4)(0, -60, -60, 0)
print("cu   njobs  fam_ms  single_ms")
res = {}
SIZES = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (64, 32, 16, 8)
MODES = tuple(sys.argv[2].split(",")) if len(sys.argv) > 2 else ("1", "0")
for fam in MODES:
    os.environ["HOP_SS_FAMILIES"] = fam
    ctx = hp.Context(W, H); L = ctx.L
    ctx.upload_orig(Y.cpu().numpy(), Cb.cpu().numpy(), Cr.cpu().numpy())
    rects = cu_rects(W, H); d_rects = torch.from_numpy(rects).to(dev)
    ctx._chk(L.hop_ssref_commit_cus_device(ctx.h, len(rects), d_rects.data_ptr(), Y.data_ptr(), Cb.data_ptr(), Cr.data_ptr()), "commit")
    cap = 1275 * wctu * hctu
    jobs = np.zeros(cap, hp.PU_JOB_DTYPE); n = 0
    for a in range(wctu * hctu):
        n += L.hop_enumerate_ctu_jobs(W, H, a, 128, pred, 2, amvp, lc, 3, 0, jobs.ctypes.data + n * jobs.itemsize, None, cap - n)
    jobs = jobs[:n]
    cu = np.maximum(jobs["w"], jobs["h"])
    for S in SIZES:
        sel = np.ascontiguousarray(jobs[cu == S])
        dj = torch.from_numpy(sel.view(np.uint8)).to(dev)
        dr = torch.zeros(len(sel) * hp.PU_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        ctx._chk(L.hop_me_search_device(ctx.h, len(sel), dj.data_ptr(), dr.data_ptr(), 1), "me"); ctx.sync()
        L.hop_profile_reset(ctx.h); L.hop_profile_enable(ctx.h, 1)
        ctx._chk(L.hop_me_search_device(ctx.h, len(sel), dj.data_ptr(), dr.data_ptr(), 1), "me")
        la, t, un = ctypes.c_uint64(), ctypes.c_double(), ctypes.c_uint64()
        L.hop_profile_read(ctx.h, 0, ctypes.byref(la), ctypes.byref(t), ctypes.byref(un))
        L.hop_profile_enable(ctx.h, 0)
        res[(S, fam)] = (len(sel), t.value, dr.cpu().numpy().tobytes())
    ctx.close()
for S in SIZES:
    a, b = res.get((S, "1")), res.get((S, "0"))
    print("%2d %7d %7.2f %7.2f %s" % (S, (a or b)[0], a[1] if a else -1, b[1] if b else -1, "" if not (a and b) else "same" if a[2] == b[2] else "DIFFERENT"))
