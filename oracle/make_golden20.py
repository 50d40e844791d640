#!/usr/bin/env python3
"""oracle/make_golden20.py -- TEST INFRASTRUCTURE.  Golden vectors of the flat (non-RDOQ) quantiser WITH sign-bit hiding from the reference's own
TComTrQuant::xQuant (oracle/ref_harness.cpp:ref_quant_flat_sbh: RDOQ switched off, the PPS's sign_data_hiding flag on, so that xQuant calls signBitHidingHDQ,
TComTrQuant.cpp:868-990): 600 blocks of every size, luma and chroma, 8 and 10 bit, I and non-I slices, the three scans (intra directions 0 / 10 / 26 on 4x4 and 8x8 luma)
-> tests/golden/quant_flat_sbh.npz; the restatement (hop_o_quant_flat_sbh) must already agree on 4000.  Needs /root/reference (build container)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import oracle


def main():
    R = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_harness.so")); R.ref_init(8, 8, 1, 1, 128)
    R.ref_quant_flat_sbh.restype = ctypes.c_uint32
    O = oracle(); O.hop_o_quant_flat_sbh.restype = ctypes.c_uint32
    rng = np.random.default_rng(20)
    par, srcs, outs, n, bad, changed = [], [], [], 0, 0, 0
    VP = ctypes.c_void_p
    while n < 4000:
        N = int(rng.choice([4, 8, 16, 32])); tt = int(rng.choice([0, 2, 3])) if N < 32 else 0
        bd = int(rng.choice([8, 10])); qp = int(rng.integers(4, 52)) if tt == 0 else int(rng.integers(4, 30)); isI = int(rng.integers(0, 2))
        intra = int(rng.integers(0, 2))
        # the scan the reference derives (getCoefScanIdx): intra luma 4x4 / 8x8 by direction, intra chroma 4x4 by the chroma direction (0 here: diagonal), otherwise diagonal
        ldir = int(rng.choice([0, 10, 26])) if (intra and tt == 0 and N <= 8) else 0
        scan = 0 if not (intra and tt == 0 and N <= 8) else (2 if 6 <= ldir <= 14 else 1 if 22 <= ldir <= 30 else 0)
        src = np.round(rng.laplace(0, 1, N * N) * rng.choice([30, 300, 3000])).astype(np.int32)
        d1 = np.zeros(N * N, np.int32); d2 = np.zeros(N * N, np.int32); d0 = np.zeros(N * N, np.int32)
        a1 = R.ref_quant_flat_sbh(src.ctypes.data_as(VP), d1.ctypes.data_as(VP), N, tt, intra, isI, qp, bd, bd, ldir)
        a2 = O.hop_o_quant_flat_sbh(bd, qp, isI, src.ctypes.data_as(VP), d2.ctypes.data_as(VP), N, scan)
        O.hop_o_quant_flat(bd, qp, isI, src.ctypes.data_as(VP), d0.ctypes.data_as(VP), N)
        bad += int(a1 != a2 or not np.array_equal(d1, d2)); n += 1; changed += int(not np.array_equal(d0, d1))
        if len(par) < 600: par.append((N, bd, qp, isI, a1, sum(len(s) for s in srcs), scan)); srcs.append(src); outs.append(d1)
    print("oracle vs reference on", n, "blocks:", bad, "mismatches;", changed, "blocks changed by the hiding")
    assert bad == 0 and changed > n // 4
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "quant_flat_sbh.npz"), par=np.array(par, np.int64), src=np.concatenate(srcs), out=np.concatenate(outs))
    print("wrote tests/golden/quant_flat_sbh.npz:", len(par), "blocks")


if __name__ == "__main__":
    main()
