#!/usr/bin/env python3
"""tools/rdoq_prof.py -- throughput of k_rdoq per TU size on synthetic coefficient batches (developer tool, GPU box)."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
from bench import _hophip
hp = _hophip()
dev = torch.device("cuda", 0)
ctx = hp.Context(64, 64); L = ctx.L
rng = np.random.default_rng(5)
tables = rng.integers(3000, 90000, (4, hp.ESTBITS_INTS)).astype(np.int32)
d_tab = torch.from_numpy(tables).to(dev)
print("size   TUs     ms    MTU/s  Mcoef/s")
for log2 in (2, 3, 4, 5):
    N2 = 1 << (2 * log2); n = (1 << 26) // N2                 # one lane per TU: the device wants >= 64k TUs of a class in flight
    yy, xx = np.mgrid[0:1 << log2, 0:1 << log2]
    src = np.round(rng.laplace(0, 1, (n, 1 << log2, 1 << log2)) * (400.0 / (1.0 + 0.35 * (xx + yy)))).astype(np.int32).reshape(-1)
    jobs = np.zeros(n, hp.RDOQ_JOB_DTYPE)
    jobs["log2_size"], jobs["comp"], jobs["scan_idx"], jobs["qp_scaled"], jobs["bit_depth"], jobs["sign_hide"], jobs["lambda"] = log2, 0, 0, 32, 8, 1, 57.9
    jobs["coeff_offset"] = np.arange(n, dtype=np.int64) * N2
    jobs["estbits_index"] = np.arange(n) % 4
    dj = torch.from_numpy(jobs.view(np.uint8)).to(dev); ds = torch.from_numpy(src).to(dev)
    dd = torch.zeros_like(ds); da = torch.zeros(n, dtype=torch.int32, device=dev)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx._chk(L.hop_rdoq_device(ctx.h, n, dj.data_ptr(), d_tab.data_ptr(), ds.data_ptr(), dd.data_ptr(), da.data_ptr()), "rdoq")
        ctx.sync(); dt = time.perf_counter() - t0
    nz = int((dd != 0).sum().item())
    print("%2dx%-2d %7d %7.2f %7.2f %8.1f   (non-zero levels %.1f%%)" % (1 << log2, 1 << log2, n, dt * 1e3, n / dt / 1e6, n * N2 / dt / 1e6, 100.0 * nz / len(src)))
