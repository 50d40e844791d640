#!/usr/bin/env python3
"""Golden vectors for rows a9/a10 (transforms, dequantiser) and a7 (intra rough search) from the REFERENCE's own
code through oracle/_ref/libref_harness.so.  TEST INFRASTRUCTURE; data only is written (tests/golden/tq.npz,
tests/golden/intra.npz).      python oracle/make_golden2.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from hoputil import ROOT, lenslet, p16, ref  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
VP = ctypes.c_void_p


def gen_tq():
    R = ref()
    rng = np.random.default_rng(41)
    blocks, fwd, invin, inv, meta = [], [], [], [], []
    for N in (4, 8, 16, 32):
        for trial in range(6):
            for bd in (8, 10):
                amp = (1 << bd) - 1
                blk = rng.integers(-amp, amp + 1, (N, N)).astype(np.int16)
                if trial % 3 == 0:
                    blk = rng.integers(-6, 7, (N, N)).astype(np.int16)
                for dst in ((0, 1) if N == 4 else (0,)):
                    a = np.zeros((N, N), np.int16)
                    R.ref_fwd_transform(bd, p16(blk), p16(a), N, N, 0 if dst else 65535)
                    co = rng.integers(-32768, 32768, (N, N)).astype(np.int16) if trial % 2 else a.copy()
                    c = np.zeros((N, N), np.int16)
                    R.ref_inv_transform(bd, p16(co), p16(c), N, N, 0 if dst else 65535)
                    meta.append([N, bd, dst]); blocks.append(blk.ravel()); fwd.append(a.ravel()); invin.append(co.ravel()); inv.append(c.ravel())
    dq_meta, dq_in, dq_out = [], [], []
    for N in (4, 8, 16, 32):
        for bd in (8, 10):
            for qp in (4, 22, 27, 32, 37, 51):
                lv = rng.integers(-40000, 40000, N * N).astype(np.int32)
                x = np.zeros(N * N, np.int32)
                R.ref_dequant_flat(bd, qp + 6 * (bd - 8), lv.ctypes.data_as(VP), x.ctypes.data_as(VP), N)
                dq_meta.append([N, bd, qp + 6 * (bd - 8)]); dq_in.append(lv); dq_out.append(x)
    np.savez_compressed(os.path.join(GOLD, "tq.npz"), meta=np.array(meta, np.int32), blocks=np.concatenate(blocks), fwd=np.concatenate(fwd),
                        invin=np.concatenate(invin), inv=np.concatenate(inv), dq_meta=np.array(dq_meta, np.int32),
                        dq_in=np.concatenate(dq_in), dq_out=np.concatenate(dq_out))
    print("tq:", len(meta), "transform cases,", len(dq_meta), "dequant cases")


def gen_intra():
    R = ref()
    rng = np.random.default_rng(43)
    W, H = 256, 192
    Y, Cb, Cr = lenslet(W, H, 15, 3)
    rec = np.ascontiguousarray(np.clip(Y + rng.integers(-4, 5, Y.shape), 0, 255).astype(np.int16))
    jobs, flags, satd = [], [], []
    for trial in range(120):
        N = (4, 8, 16, 32, 64)[trial % 5]
        x = int(rng.integers(1, (W - 2 * N) // 4)) * 4
        y = int(rng.integers(1, (H - 2 * N) // 4)) * 4
        U = N // 4
        units = 4 * U + 1
        kind = trial % 6
        fl = np.ones(68, np.uint8)
        fl[units:] = 0
        if kind == 1:
            fl[:] = 0
        elif kind == 2:
            fl[:U] = 0
        elif kind == 3:
            fl[:2 * U + 1] = 0
        elif kind == 4:
            fl[:units] = rng.integers(0, 2, units)
        elif kind == 5:
            fl[3 * U + 1 + int(rng.integers(0, U)):units] = 0
            fl[:int(rng.integers(0, U))] = 0
        strong = trial % 2
        s1 = (ctypes.c_uint32 * 35)()
        R.ref_intra_rough(p16(rec), W, p16(Y), W, x, y, N, fl.ctypes.data_as(VP), 8, strong, s1, -1, None, None)
        jobs.append([x, y, N, strong]); flags.append(fl); satd.append(list(s1))
    np.savez_compressed(os.path.join(GOLD, "intra.npz"), Y=Y.astype(np.uint8), rec=rec.astype(np.uint8), jobs=np.array(jobs, np.int32),
                        flags=np.array(flags, np.uint8), satd=np.array(satd, np.uint32))
    print("intra:", len(jobs), "blocks")


if __name__ == "__main__":
    gen_tq()
    gen_intra()
