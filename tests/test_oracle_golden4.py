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


def tu_rd_cases():
    g = load("tu_rd.npz")
    for i in range(len(g["par"])):
        log2, comp, qp, trd, sh, uts, slot, off = (int(v) for v in g["par"][i])
        n = 1 << (2 * log2)
        yield dict(log2=log2, comp=comp, qp=qp, trd=trd, sh=sh, uts=uts, slot=slot, lamq=float(g["lam"][i][0]), lam=float(g["lam"][i][1]), w=float(g["lam"][i][2]),
                   st=np.ascontiguousarray(g["st"][i]), resi=np.ascontiguousarray(g["resi"][off:off + n], np.int16), out=g["out"][i], cost=float(g["cost"][i]),
                   levels=np.ascontiguousarray(g["levels"][off:off + n], np.int32))


def test_tu_rd_oracle_vs_golden():
    """row a8b leaf step: the composite restatement against the vectors assembled from the reference's own members"""
    O = oracle()
    VP = ctypes.c_void_p
    O.hop_o_tu_rd.argtypes = [VP] + [ctypes.c_int] * 7 + [ctypes.c_double] * 3 + [VP, ctypes.c_uint32, VP, VP, VP]
    n = coded = 0
    for c in tu_rd_cases():
        lv = np.zeros(len(c["resi"]), np.int32); o = np.zeros(8, np.uint32); cost = ctypes.c_double()
        fl = int(c["st"][150]) | (int(c["st"][151]) << 8)
        assert O.hop_o_tu_rd(c["resi"].ctypes.data, c["log2"], c["comp"], c["qp"], 8, c["trd"], c["sh"], c["uts"], c["lamq"], c["lam"], c["w"],
                             c["st"].ctypes.data, fl, lv.ctypes.data, o.ctypes.data, ctypes.byref(cost)) == 0
        assert np.array_equal(o, c["out"]) and cost.value == c["cost"] and np.array_equal(lv, c["levels"]), (n, c["log2"], c["comp"])
        n += 1; coded += int(o[0] != 0)
    assert n == 192 and coded > 60
