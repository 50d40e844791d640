"""CABAC bit estimator of residual coding (building block of rows a0 / a8 / a8b / a11): the CPU restatement against golden
vectors made by the reference's own TEncSbac + TEncBinCABACCounter (oracle/make_golden4.py)."""
import ctypes
import os
import sys

import numpy as np

from goldutil import load
from hoputil import ROOT, oracle

sys.path.insert(0, os.path.join(ROOT, "oracle"))


def cabac_golden():
    return load("cabac.npz")


def test_cabac_oracle_vs_golden():
    import make_golden4 as mg
    O = oracle(); mg.bind(O, None)
    g = cabac_golden()
    for st in range(5):
        for qp in range(52):
            b = np.zeros(150, np.uint8)
            assert O.hop_o_cabac_init(b.ctypes.data, st, qp) == 0
            assert np.array_equal(b, g["init"][st, qp]), (st, qp)
    for st, (w, comp), want in zip(g["est_states"], g["est_par"], g["est_out"]):
        e = np.full(244, 0x5A5A, np.int32)
        O.hop_o_cabac_est_bits(np.ascontiguousarray(st).ctypes.data, int(w), int(comp), e.ctypes.data)
        assert np.array_equal(e, want), (w, comp)
    par, coef, bits = g["par"], g["coef"], g["bits"]
    for (sl, qp, t0, t1), final in zip(g["chains"], g["finals"]):
        s = g["init"][sl, qp].copy()
        for t in range(t0, t1):
            log2, comp, scan, sh, uts, tsf, off = (int(v) for v in par[t])
            c = np.ascontiguousarray(coef[off:off + (1 << (2 * log2))], np.int32)
            f = O.hop_o_cabac_coeff_bits(s.ctypes.data, c.ctypes.data, log2, comp, scan, sh, uts, tsf)
            assert f == int(bits[t]), (t, log2, comp, scan)
        assert np.array_equal(s, final)


def test_cabac_tables_consistent():
    """state transitions stay inside the table, the MPS path lowers the cost of the MPS, a flag costs about one bit at the equiprobable state"""
    O = oracle()
    O.hop_o_ctx_bits.restype = ctypes.c_int32; O.hop_o_ctx_next.restype = ctypes.c_uint8
    for s in range(126):
        for b in (0, 1):
            assert 0 <= O.hop_o_ctx_next(s, b) < 128
        mps = s & 1
        assert O.hop_o_ctx_bits(O.hop_o_ctx_next(s, mps), mps) <= O.hop_o_ctx_bits(s, mps)
    assert abs(O.hop_o_ctx_bits(0, 0) - 32768) < 2048 and abs(O.hop_o_ctx_bits(0, 1) - 32768) < 2048
