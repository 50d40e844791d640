#!/usr/bin/env python3
"""oracle/make_golden5.py -- golden vectors for the leaf step of the residual quadtree (row a8b): one component TU through the
reference's own estBit / transformNxN (xT + xRateDistOptQuant) / codeQtCbf + codeCoeffNxN with the counting coder / invtransformNxN /
getDistPart / calcRdCost, sequenced as TEncSearch::xEstimateResidualQT does (oracle/ref_harness.cpp:ref_tu_rd).
Build container only; writes tests/golden/tu_rd.npz (192 TUs laid out in one 256x256 8-bit picture)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import oracle, ref
VP = ctypes.c_void_p


def bind(O, R):
    O.hop_o_tu_rd.argtypes = [VP] + [ctypes.c_int] * 7 + [ctypes.c_double] * 3 + [VP, ctypes.c_uint32, VP, VP, VP]
    if R is not None:
        R.ref_cabac_init.argtypes = [ctypes.c_int, ctypes.c_int, VP]
        R.ref_tu_rd.argtypes = [VP] + [ctypes.c_int] * 7 + [ctypes.c_double] * 3 + [VP, ctypes.c_uint, VP, VP, VP]


def case(rng, R, comp, bd=8):
    log2 = int(rng.integers(2, 6 if comp == 0 else 5)); N = 1 << log2
    qp = int(rng.integers(18, 44)); trd = int(rng.integers(0, 3)); sh = int(rng.integers(0, 2)); uts = int(rng.integers(0, 2))
    lam = 0.57 * 2.0 ** ((qp - 12) / 3.0) * float(rng.uniform(0.6, 1.8))
    w = 1.0 if comp == 0 else float(rng.uniform(0.7, 1.3))
    st = np.zeros(152, np.uint8); R.ref_cabac_init(int(rng.integers(0, 5)), int(rng.integers(20, 45)), st.ctypes.data)
    st[:150] = np.clip(st[:150].astype(int) + rng.integers(-6, 7, 150), 0, 125)
    fl = int(rng.integers(0, 32768)); st[150] = fl & 255; st[151] = fl >> 8
    amp = float(rng.choice([0.7, 1.5, 3.0, 8.0, 25.0, 60.0]))
    yy, xx = np.mgrid[0:N, 0:N]
    resi = (rng.normal(0, amp, (N, N)) + amp * np.sin(xx * rng.uniform(0, 1) + yy * rng.uniform(0, 1))).round()
    resi = np.clip(resi, -120, 120).astype(np.int16)                # realisable as original - 128 in an 8-bit picture
    return dict(log2=log2, comp=comp, qp=qp, trd=trd, sh=sh, uts=uts, lam=lam, lamq=lam if comp == 0 else lam / w, w=w, st=st, fl=fl,
                resi=np.ascontiguousarray(resi.reshape(-1)), bd=bd)


def run(fn, c, use_log2):
    N = 1 << c["log2"]
    lv = np.zeros(N * N, np.int32); o = np.zeros(8, np.uint32); cost = ctypes.c_double()
    fn(c["resi"].ctypes.data, c["log2"] if use_log2 else N, c["comp"] if use_log2 else (0, 2, 3)[c["comp"]], c["qp"], c["bd"], c["trd"], c["sh"], c["uts"],
       c["lamq"], c["lam"], c["w"], c["st"].ctypes.data, c["fl"], lv.ctypes.data, o.ctypes.data, ctypes.byref(cost))
    return lv, o, cost.value


def main():
    O, R = oracle(), ref()
    bind(O, R)
    rng = np.random.default_rng(123)
    bad = coded = nulls = 0
    for i in range(3000):                                            # the restatement against the reference pieces, bulk
        c = case(rng, R, int(rng.integers(0, 3)), bd=int(rng.choice([8, 10])))
        a, b = run(R.ref_tu_rd, c, False), run(O.hop_o_tu_rd, c, True)
        bad += int(not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]))
        coded += int(a[1][0] != 0); nulls += int(a[1][4] != 0 and a[1][0] == 0)
    print("oracle vs reference pieces: 3000 TUs, %d coded, %d cbf-zero decisions, %d mismatches" % (coded, nulls, bad))
    assert bad == 0
    rng = np.random.default_rng(124)
    par, lam, st, resi, outs, costs, levels = [], [], [], [], [], [], []
    for comp in (0, 1, 2):
        for k in range(64):
            c = case(rng, R, comp)
            lv, o, cost = run(R.ref_tu_rd, c, False)
            par.append([c["log2"], comp, c["qp"], c["trd"], c["sh"], c["uts"], k, sum(len(r) for r in resi)])
            lam.append([c["lamq"], c["lam"], c["w"]]); st.append(c["st"]); resi.append(c["resi"]); outs.append(o); costs.append(cost); levels.append(lv)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tu_rd.npz"), par=np.array(par, np.int64), lam=np.array(lam, np.float64), st=np.stack(st),
                        resi=np.concatenate(resi), out=np.stack(outs), cost=np.array(costs, np.float64), levels=np.concatenate(levels))
    print("wrote tests/golden/tu_rd.npz:", len(par), "TUs")


if __name__ == "__main__":
    main()
