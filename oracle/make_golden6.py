#!/usr/bin/env python3
"""oracle/make_golden6.py -- golden vectors for the intra leaf step (row a8: TEncSearch::xIntraCodingLumaBlk / ChromaBlk after the
prediction) from the reference's own members (oracle/ref_harness.cpp:ref_tu_intra).  Build container only; writes
tests/golden/tu_intra.npz (192 TUs laid out in one 256x256 8-bit picture: original and prediction planes + expectations)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import oracle, ref
VP = ctypes.c_void_p


def main():
    O, R = oracle(), ref()
    R.ref_cabac_init.argtypes = [ctypes.c_int, ctypes.c_int, VP]
    R.ref_tu_intra.argtypes = [VP, VP] + [ctypes.c_int] * 9 + [ctypes.c_double] * 3 + [VP, ctypes.c_uint, VP, VP, VP, VP]
    O.hop_o_tu_intra.argtypes = [VP, VP] + [ctypes.c_int] * 9 + [ctypes.c_double] * 3 + [VP, ctypes.c_uint32, VP, VP, VP, VP]
    O.hop_o_coef_scan_idx.argtypes = [ctypes.c_int] * 4
    rng = np.random.default_rng(606)
    W = H = 256
    org = [np.zeros((H, W), np.int16), np.zeros((H // 2, W // 2), np.int16), np.zeros((H // 2, W // 2), np.int16)]
    prd = [a.copy() for a in org]; rec = [a.copy() for a in org]
    par, lams, sts, outs, costs, levels = [], [], [], [], [], []
    for comp in (0, 1, 2):
        for k in range(64):
            log2 = int(rng.integers(2, 6 if comp == 0 else 5)); N = 1 << log2
            qp = int(rng.integers(18, 44)); trd = int(rng.integers(0, 3)); sh = int(rng.integers(0, 2)); uts = int(rng.integers(0, 2))
            ldir = int(rng.integers(0, 35)); cdir = int(rng.choice([0, 1, 10, 26, 34]))
            lam = 0.57 * 2.0 ** ((qp - 12) / 3.0) * float(rng.uniform(0.6, 1.8)); w = 1.0 if comp == 0 else float(rng.uniform(0.7, 1.3))
            st = np.zeros(152, np.uint8); R.ref_cabac_init(int(rng.integers(0, 5)), int(rng.integers(20, 45)), st.ctypes.data)
            st[:150] = np.clip(st[:150].astype(int) + rng.integers(-6, 7, 150), 0, 125)
            fl = int(rng.integers(0, 32768)); st[150] = fl & 255; st[151] = fl >> 8
            base = int(rng.integers(0, 256)); amp = float(rng.choice([1.0, 4.0, 15.0, 60.0]))
            p = np.clip(base + rng.normal(0, amp, (N, N)), 0, 255).round().astype(np.int16)
            o = np.clip(p + rng.normal(0, amp, (N, N)), 0, 255).round().astype(np.int16)
            if rng.random() < 0.1: o[:] = int(rng.choice([0, 255]))
            scan = O.hop_o_coef_scan_idx(N, int(comp == 0), 1, ldir if comp == 0 else cdir)
            pf = np.ascontiguousarray(p.reshape(-1)); of = np.ascontiguousarray(o.reshape(-1))
            lv = np.zeros(N * N, np.int32); rc = np.zeros(N * N, np.int16); ou = np.zeros(8, np.uint32); cost = ctypes.c_double()
            R.ref_tu_intra(of.ctypes.data, pf.ctypes.data, N, (0, 2, 3)[comp], ldir, cdir, qp, 8, trd, sh, uts, lam if comp == 0 else lam / w, lam, w, st.ctypes.data, fl,
                           lv.ctypes.data, rc.ctypes.data, ou.ctypes.data, ctypes.byref(cost))
            lv2 = np.zeros(N * N, np.int32); rc2 = np.zeros(N * N, np.int16); ou2 = np.zeros(8, np.uint32); cost2 = ctypes.c_double()
            O.hop_o_tu_intra(of.ctypes.data, pf.ctypes.data, log2, comp, scan, 1, qp, 8, trd, sh, uts, lam if comp == 0 else lam / w, lam, w, st.ctypes.data, fl,
                             lv2.ctypes.data, rc2.ctypes.data, ou2.ctypes.data, ctypes.byref(cost2))
            assert np.array_equal(lv, lv2) and np.array_equal(rc, rc2) and np.array_equal(ou, ou2) and cost.value == cost2.value
            px, py = (32 * (k % 8), 32 * (k // 8)) if comp == 0 else (16 * (k % 8), 16 * (k // 8))
            org[comp][py:py + N, px:px + N] = o; prd[comp][py:py + N, px:px + N] = p; rec[comp][py:py + N, px:px + N] = rc.reshape(N, N)
            par.append([log2, comp, qp, trd, sh, uts, scan, px, py, sum(len(l) for l in levels)]); lams.append([lam if comp == 0 else lam / w, lam, w])
            sts.append(st); outs.append(ou); costs.append(cost.value); levels.append(lv)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tu_intra.npz"), par=np.array(par, np.int64), lam=np.array(lams), st=np.stack(sts), out=np.stack(outs),
                        cost=np.array(costs), levels=np.concatenate(levels), orgY=org[0], orgCb=org[1], orgCr=org[2], prdY=prd[0], prdCb=prd[1], prdCr=prd[2],
                        recY=rec[0], recCb=rec[1], recCr=rec[2])
    print("wrote tests/golden/tu_intra.npz:", len(par), "TUs, coded", int(sum(o[0] != 0 for o in outs)))


if __name__ == "__main__":
    main()
