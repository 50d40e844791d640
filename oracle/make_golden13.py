#!/usr/bin/env python3
"""oracle/make_golden13.py -- TEST INFRASTRUCTURE.  Golden vectors of the flat (non-RDOQ) quantiser from the reference's own TComTrQuant::xQuant
(oracle/ref_harness.cpp:ref_quant_flat: RDOQ switched off, a CU at the slice QP, sign-bit hiding off): 400 blocks of every size, luma and chroma,
8 and 10 bit, I and non-I slices (the rounding offset differs) -> tests/golden/quant_flat.npz; the restatement must already agree on 3000.
Needs /root/reference (build container)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import oracle


def main():
    R = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_harness.so")); R.ref_init(8, 8, 1, 1, 128)
    R.ref_quant_flat.restype = ctypes.c_uint32
    O = oracle(); O.hop_o_quant_flat.restype = ctypes.c_uint32
    rng = np.random.default_rng(13)
    par, srcs, outs, n, bad = [], [], [], 0, 0
    while n < 3000:
        N = int(rng.choice([4, 8, 16, 32])); tt = int(rng.choice([0, 2, 3])) if N < 32 else 0
        bd = int(rng.choice([8, 10])); qp = int(rng.integers(4, 52)) if tt == 0 else int(rng.integers(4, 30)); isI = int(rng.integers(0, 2))
        src = np.round(rng.laplace(0, 1, N * N) * rng.choice([30, 300, 3000])).astype(np.int32)
        d1 = np.zeros(N * N, np.int32); d2 = np.zeros(N * N, np.int32)
        a1 = R.ref_quant_flat(src.ctypes.data_as(ctypes.c_void_p), d1.ctypes.data_as(ctypes.c_void_p), N, tt, int(rng.integers(0, 2)), isI, qp, bd, bd)
        a2 = O.hop_o_quant_flat(bd, qp, isI, src.ctypes.data_as(ctypes.c_void_p), d2.ctypes.data_as(ctypes.c_void_p), N)
        bad += int(a1 != a2 or not np.array_equal(d1, d2)); n += 1
        if len(par) < 400: par.append((N, bd, qp, isI, a1, sum(len(s) for s in srcs))); srcs.append(src); outs.append(d1)
    print("oracle vs reference on", n, "blocks:", bad, "mismatches")
    assert bad == 0
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "quant_flat.npz"), par=np.array(par, np.int64), src=np.concatenate(srcs), out=np.concatenate(outs))
    print("wrote tests/golden/quant_flat.npz:", len(par), "blocks")


if __name__ == "__main__":
    main()
