"""CPU: oracle transforms / dequantiser / intra rough search against the golden vectors produced by the REFERENCE's
own xTrMxN, xITrMxN, xDeQuant, fillReferenceSamples, predIntraLumaAng and calcHAD (oracle/make_golden2.py).
The forward flat quantiser is checked against vectors of the reference's xQuant with RDOQ off (oracle/make_golden13.py) and through the
round-trip property below."""
import ctypes

import numpy as np

from goldutil import load
from hoputil import oracle, p16

VP = ctypes.c_void_p


def test_transforms_and_dequant_golden():
    O = oracle()
    g = load("tq.npz")
    off = 0
    for (N, bd, dst) in g["meta"]:
        N, bd, dst = int(N), int(bd), int(dst)
        n2 = N * N
        blk = np.ascontiguousarray(g["blocks"][off:off + n2]); want_f = g["fwd"][off:off + n2]
        co = np.ascontiguousarray(g["invin"][off:off + n2]); want_i = g["inv"][off:off + n2]
        a = np.zeros(n2, np.int16); b = np.zeros(n2, np.int16)
        O.hop_o_fwd_transform(bd, p16(blk), p16(a), N, dst)
        O.hop_o_inv_transform(bd, p16(co), p16(b), N, dst)
        assert np.array_equal(a, want_f) and np.array_equal(b, want_i), (N, bd, dst)
        off += n2
    off = 0
    for (N, bd, qps) in g["dq_meta"]:
        n2 = int(N) * int(N)
        lv = np.ascontiguousarray(g["dq_in"][off:off + n2]); y = np.zeros(n2, np.int32)
        O.hop_o_dequant_flat(int(bd), int(qps), lv.ctypes.data_as(VP), y.ctypes.data_as(VP), int(N))
        assert np.array_equal(y, g["dq_out"][off:off + n2])
        off += n2


def test_flat_quantiser_properties():
    """sign symmetry, monotonicity in |coef|, and |dequant(quant(c)) - c| bounded by one step"""
    O = oracle()
    O.hop_o_quant_flat.restype = ctypes.c_uint32
    rng = np.random.default_rng(3)
    for N in (4, 8, 16, 32):
        for qp in (22, 32, 37):
            c = rng.integers(-20000, 20000, N * N).astype(np.int32)
            l1 = np.zeros(N * N, np.int32); l2 = np.zeros(N * N, np.int32); d = np.zeros(N * N, np.int32)
            s1 = O.hop_o_quant_flat(8, qp, 0, c.ctypes.data_as(VP), l1.ctypes.data_as(VP), N)
            O.hop_o_quant_flat(8, qp, 0, (-c).ctypes.data_as(VP), l2.ctypes.data_as(VP), N)
            assert np.array_equal(l1, -l2) and s1 == int(np.abs(l1).sum())
            order = np.argsort(np.abs(c)); assert (np.diff(np.abs(l1)[order]) >= 0).all()
            O.hop_o_dequant_flat(8, qp, l1.ctypes.data_as(VP), d.ctypes.data_as(VP), N)
            log2N = int(np.log2(N)); step = (2.0 ** ((qp - 4) / 6.0)) * (1 << (15 - 8 - log2N)) / 64.0 * 64.0
            ok = np.abs(l1) < 32767
            assert (np.abs(d[ok] - np.clip(c[ok], -32768, 32767)) <= step * 1.05 + 1).all()


def test_intra_rough_golden():
    O = oracle()
    g = load("intra.npz")
    Y = g["Y"].astype(np.int16); rec = np.ascontiguousarray(g["rec"].astype(np.int16))
    W = Y.shape[1]
    for (x, y, N, strong), fl, want in zip(g["jobs"], g["flags"], g["satd"]):
        s = (ctypes.c_uint32 * 35)()
        fl = np.ascontiguousarray(fl)
        O.hop_o_intra_rough(p16(rec), W, p16(Y), W, int(x), int(y), int(N), fl.ctypes.data_as(VP), 8, int(strong), s)
        assert list(s) == [int(v) for v in want], (int(x), int(y), int(N))


def test_flat_quantiser_golden():
    """row a10, forward side: hop_o_quant_flat against 400 blocks quantised by the reference's own xQuant with RDOQ off (tests/golden/quant_flat.npz,
    oracle/make_golden13.py): every size, 8 / 10 bit, I and non-I rounding offsets"""
    import ctypes
    g = load("quant_flat.npz")
    O = oracle(); O.hop_o_quant_flat.restype = ctypes.c_uint32
    for N, bd, qp, isI, asum, off in g["par"]:
        N, off = int(N), int(off)
        src = np.ascontiguousarray(g["src"][off:off + N * N], np.int32); d = np.zeros(N * N, np.int32)
        a = O.hop_o_quant_flat(int(bd), int(qp), int(isI), src.ctypes.data_as(ctypes.c_void_p), d.ctypes.data_as(ctypes.c_void_p), N)
        assert a == int(asum) and np.array_equal(d, g["out"][off:off + N * N]), (N, bd, qp, isI)


def test_flat_quantiser_sign_bit_hiding_golden():
    """row a10 with the PPS's sign_data_hiding flag: hop_o_quant_flat_sbh (flat quantiser, remainders, TComTrQuant::signBitHidingHDQ along the TU's scan) against 600 blocks
    quantised by the reference's own xQuant with RDOQ off and sign hiding on (tests/golden/quant_flat_sbh.npz, oracle/make_golden20.py: every size, 8 / 10 bit, the three
    scans); unreachable in the shipped configurations (they quantise with RDOQ, row a11, which has its own hiding) but part of the row"""
    import ctypes
    g = load("quant_flat_sbh.npz")
    O = oracle(); O.hop_o_quant_flat_sbh.restype = ctypes.c_uint32; O.hop_o_quant_flat.restype = ctypes.c_uint32
    changed = 0
    for N, bd, qp, isI, asum, off, scan in g["par"]:
        N, off = int(N), int(off)
        src = np.ascontiguousarray(g["src"][off:off + N * N], np.int32); d = np.zeros(N * N, np.int32); d0 = np.zeros(N * N, np.int32)
        a = O.hop_o_quant_flat_sbh(int(bd), int(qp), int(isI), src.ctypes.data_as(ctypes.c_void_p), d.ctypes.data_as(ctypes.c_void_p), N, int(scan))
        assert a == int(asum) and np.array_equal(d, g["out"][off:off + N * N]), (N, bd, qp, isI, scan)
        O.hop_o_quant_flat(int(bd), int(qp), int(isI), src.ctypes.data_as(ctypes.c_void_p), d0.ctypes.data_as(ctypes.c_void_p), N)
        changed += int(not np.array_equal(d, d0))
    assert changed > 300
