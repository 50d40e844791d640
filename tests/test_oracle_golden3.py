"""Row a11 (RDOQ): the CPU restatement against golden vectors produced by the reference's own
TComTrQuant::xRateDistOptQuant (oracle/make_golden3.py), and -- where the reference harness exists (build container) --
against the reference function itself on fresh seeded cases."""
import ctypes
import os
import sys

import numpy as np
import pytest

from goldutil import load
from hoputil import ROOT, oracle, ref_available

sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _bind(O):
    O.hop_o_rdoq.restype = ctypes.c_int
    O.hop_o_rdoq.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 8 + [ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]


def rdoq_cases():
    g = load("rdoq.npz")
    par, lam, src, eb, out = g["par"], g["lam"], g["src"], g["eb"], g["out"]
    for i in range(len(par)):
        log2, comp, intra, scan, tr, qp, bd, sh, asum, off = (int(v) for v in par[i])
        n = 1 << (2 * log2)
        yield dict(log2=log2, comp=comp, intra=intra, scan=scan, tr=tr, qp=qp, bd=bd, sh=sh, lam=float(lam[i]), asum=asum,
                   src=np.ascontiguousarray(src[off:off + n], np.int32), eb=np.ascontiguousarray(eb[i], np.int32),
                   out=np.ascontiguousarray(out[off:off + n], np.int32))


def test_rdoq_oracle_vs_golden():
    O = oracle(); _bind(O)
    n = nz = 0
    for c in rdoq_cases():
        d = np.zeros(len(c["src"]), np.int32); a = ctypes.c_uint32(0)
        assert O.hop_o_rdoq(c["src"].ctypes.data, d.ctypes.data, c["log2"], c["comp"], c["intra"], c["scan"], c["tr"], c["qp"], c["bd"], c["sh"],
                            c["lam"], c["eb"].ctypes.data, ctypes.byref(a)) == 0
        assert a.value == c["asum"] and np.array_equal(d, c["out"]), (n, c["log2"], c["comp"], c["intra"], c["scan"], c["qp"], c["sh"])
        n += 1; nz += int(np.any(d))
    assert n == 360 and nz > 250


def test_scan_tables():
    """every scan is a permutation; diagonal 4x4 as in the standard; coefficient-group scans cover every group"""
    O = oracle()
    O.hop_o_scan.restype = ctypes.POINTER(ctypes.c_uint32); O.hop_o_scan_cg.restype = ctypes.POINTER(ctypes.c_uint32)
    for s in range(3):
        for l2 in range(2, 6):
            n = 1 << (2 * l2)
            assert sorted(O.hop_o_scan(s, l2)[:n]) == list(range(n))
            assert sorted(O.hop_o_scan_cg(s, l2)[:n // 16]) == list(range(n // 16))
    assert O.hop_o_scan(0, 2)[:16] == [0, 4, 1, 8, 5, 2, 12, 9, 6, 3, 13, 10, 7, 14, 11, 15]


@pytest.mark.skipif(not ref_available(), reason="reference harness only exists in the build container")
def test_rdoq_oracle_vs_reference_function():
    import make_golden3 as mg
    from hoputil import ref
    O, R = oracle(), ref()
    mg.bind(O, R)
    rng = np.random.default_rng(4711)
    for i in range(600):
        c = mg.case(rng, i % 2 == 0)
        d, a, scan = mg.run_ref(R, c)
        d2, a2 = mg.run_oracle(O, c, scan)
        assert a == a2 and np.array_equal(d, d2), (i, c["log2"], c["comp"], c["qp"])
