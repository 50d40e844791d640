#!/usr/bin/env python3
"""oracle/make_golden4.py -- golden vectors for the CABAC bit estimator of residual coding (context initialisation,
TEncSbac::estBit, counted bits of codeCoeffNxN with context update) from the reference's own TEncSbac + TEncBinCABACCounter
(oracle/_ref/libref_harness.so).  Build container only; writes tests/golden/cabac.npz."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import oracle, ref

EBI = 244


def bind(O, R):
    O.hop_o_cabac_coeff_bits.restype = ctypes.c_uint64
    O.hop_o_cabac_coeff_bits.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6
    O.hop_o_cabac_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    O.hop_o_cabac_est_bits.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    O.hop_o_coef_scan_idx.argtypes = [ctypes.c_int] * 4
    if R is not None:
        R.ref_cabac_coeff_bits.restype = ctypes.c_uint64
        R.ref_cabac_coeff_bits.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 8
        R.ref_cabac_init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        R.ref_cabac_est_bits.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]


def tu(rng):
    log2 = int(rng.integers(2, 6)); comp = int(rng.integers(0, 3))
    if comp and log2 == 5: log2 = 4
    N = 1 << log2
    intra = int(rng.integers(0, 2)); ldir = int(rng.integers(0, 35)); cdir = int(rng.choice([0, 1, 10, 26, 34]))
    sh = int(rng.integers(0, 2)); uts = int(rng.integers(0, 2)); tsf = int(rng.integers(0, 2))
    dens = float(rng.choice([0.02, 0.1, 0.3, 0.8])); mag = float(rng.choice([1.0, 3.0, 20.0, 500.0]))
    yy, xx = np.mgrid[0:N, 0:N]
    c = np.round(rng.laplace(0, 1, (N, N)) * mag / (1 + 0.3 * (xx + yy))).astype(np.int32)
    c[rng.random((N, N)) > dens] = 0
    if rng.random() < 0.05: c[:] = 0
    return dict(log2=log2, comp=comp, intra=intra, ldir=ldir, cdir=cdir, sh=sh, uts=uts, tsf=tsf, coef=np.ascontiguousarray(c.reshape(-1)))


def main():
    O, R = oracle(), ref()
    bind(O, R)
    # initialisation: every slice type x QP
    init = np.zeros((5, 52, 150), np.uint8)
    for st in range(5):
        for qp in range(52):
            R.ref_cabac_init(st, qp, init[st, qp].ctypes.data)
            b = np.zeros(150, np.uint8); O.hop_o_cabac_init(b.ctypes.data, st, qp)
            assert np.array_equal(b, init[st, qp])
    # the CU-level sets of an SS/GT CU's syntax (hop_cabac_cu_ctx): every slice type x QP -> tests/golden/cabac_cu.npz
    cu_init = np.zeros((5, 52, 19), np.uint8)
    R.ref_cabac_cu_init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    for st in range(5):
        for qp in range(52):
            R.ref_cabac_cu_init(st, qp, cu_init[st, qp].ctypes.data)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cabac_cu.npz"), init=cu_init)
    rng = np.random.default_rng(99)
    # estBit from random states
    est_states, est_par, est_out = [], [], []
    for i in range(120):
        st = rng.integers(0, 128, 150).astype(np.uint8)
        w = int(rng.choice([4, 8, 16, 32])); comp = int(rng.integers(0, 3))
        if comp and w == 32: w = 16
        e = np.full(EBI, 0x5A5A, np.int32); e2 = e.copy()
        R.ref_cabac_est_bits(st.ctypes.data, w, (0, 2, 3)[comp], e.ctypes.data)
        O.hop_o_cabac_est_bits(st.ctypes.data, w, comp, e2.ctypes.data)
        assert np.array_equal(e, e2)
        est_states.append(st); est_par.append([w, comp]); est_out.append(e)
    # chains of TUs from an initial state: bits per TU and the final states
    chains, par, coefs, bits, finals = [], [], [], [], []
    nbad = 0
    for i in range(4000):
        sl = int(rng.integers(0, 5)); qp = int(rng.integers(0, 52))
        s1 = init[sl, qp].copy(); s2 = s1.copy()
        keep = i < 150
        if keep: chains.append([sl, qp, len(par)])
        for k in range(int(rng.integers(1, 6))):
            t = tu(rng); N = 1 << t["log2"]
            scan = O.hop_o_coef_scan_idx(N, int(t["comp"] == 0), t["intra"], t["ldir"] if t["comp"] == 0 else t["cdir"])
            f1 = R.ref_cabac_coeff_bits(s1.ctypes.data, t["coef"].ctypes.data, N, (0, 2, 3)[t["comp"]], t["intra"], t["ldir"], t["cdir"], t["sh"], t["uts"], t["tsf"])
            f2 = O.hop_o_cabac_coeff_bits(s2.ctypes.data, t["coef"].ctypes.data, t["log2"], t["comp"], scan, t["sh"], t["uts"], t["tsf"])
            nbad += int(f1 != f2 or not np.array_equal(s1, s2))
            if keep:
                par.append([t["log2"], t["comp"], scan, t["sh"], t["uts"], t["tsf"], sum(len(c) for c in coefs)]); coefs.append(t["coef"]); bits.append(f1)
        if keep: chains[-1].append(len(par)); finals.append(s1.copy())
    print("oracle vs reference on 4000 chains:", nbad, "mismatches")
    assert nbad == 0
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cabac.npz"), init=init, est_states=np.stack(est_states), est_par=np.array(est_par, np.int32),
                        est_out=np.stack(est_out), chains=np.array(chains, np.int64), par=np.array(par, np.int64), coef=np.concatenate(coefs),
                        bits=np.array(bits, np.uint64), finals=np.stack(finals))
    print("wrote tests/golden/cabac.npz:", len(chains), "chains,", len(par), "TUs")


if __name__ == "__main__":
    main()
