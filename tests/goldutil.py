"""Rebuild the inputs of the golden fixtures (tests/golden/*.npz) -- shared by the CPU and GPU tests."""
import ctypes
import os
import zlib

import numpy as np

from hoputil import ROOT, Planes, oracle, p16

GOLD = os.path.join(ROOT, "tests", "golden")


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def load(name):
    return np.load(os.path.join(GOLD, name))


def coded_planes(Y, Cb, Cr, W, H, ctu_r, ctu_c, partial):
    """Same construction as oracle/make_golden.py:coded_planes, with the oracle's border extension."""
    O = oracle()
    pl = Planes(W, H)

    def put(x, y, sx, sy):
        pl.y00()[y:y + sy, x:x + sx] = Y[y:y + sy, x:x + sx]
        pl.bufCb[40 + y // 2:40 + (y + sy) // 2, 40 + x // 2:40 + (x + sx) // 2] = Cb[y // 2:(y + sy) // 2, x // 2:(x + sx) // 2]
        pl.bufCr[40 + y // 2:40 + (y + sy) // 2, 40 + x // 2:40 + (x + sx) // 2] = Cr[y // 2:(y + sy) // 2, x // 2:(x + sx) // 2]
    if ctu_r > 0:
        put(0, 0, W, ctu_r * 64)
    if ctu_c > 0:
        put(0, ctu_r * 64, ctu_c * 64, min(64, H - ctu_r * 64))
    for (x, y, s) in partial:
        put(int(x), int(y), int(s), int(s))
    # border extension only: re-commit the 8x8 block at (0,0) with the values it already holds
    by = np.ascontiguousarray(pl.y00()[0:8, 0:8])
    bb = np.ascontiguousarray(pl.bufCb[40:44, 40:44])
    br = np.ascontiguousarray(pl.bufCr[40:44, 40:44])
    O.hop_o_ssref_commit_cu(pl.ptr00(0), pl.ptr00(1), pl.ptr00(2), W, H, 0, 0, 8, p16(by), p16(bb), p16(br))
    return pl


def me_chain_scenarios():
    """Yields (planes, Y, jobs, outs, lambda_cost) per scenario of me_chain.npz."""
    g = load("me_chain.npz")
    W, H = int(g["W"]), int(g["H"])
    Y, Cb, Cr, rec = (g[k].astype(np.int16) for k in ("Y", "Cb", "Cr", "rec"))
    partial = g["partial"]
    pi = 0
    for si, (cr_, cc_, npart) in enumerate(g["scen"]):
        part = [tuple(int(v) for v in partial[pi + k]) for k in range(npart)]
        pi += npart
        pl = coded_planes(rec, Cb, Cr, W, H, int(cr_), int(cc_), part)
        assert [crc(pl.bufY), crc(pl.bufCb), crc(pl.bufCr)] == [int(v) for v in g["planes_crc"][si]]
        sel = g["jobs"][:, 0] == si
        yield pl, Y, g["jobs"][sel], g["outs"][sel], int(g["lambda_cost"])


def encoder_calls():
    """tests/golden/encoder_calls.npz (oracle/make_golden7.py): the PUs sampled from a real encode of the 128x128 golden lenslet.
    Yields (planes, Y, meta): the SS-reference luma plane holding exactly the samples the reference's members could read for this
    PU (everything else the -1 sentinel; the padded buffer's first sample set valid, as it is once a CU has been committed,
    TEncSearch.cpp:4605), the frame's original luma and the 44 recorded numbers (layout in make_golden7.py)."""
    from hoputil import MARGIN_Y, lenslet
    g = load("encoder_calls.npz")
    W, H = int(g["W"]), int(g["H"])
    Y, _, _ = lenslet(W, H, 16, int(g["seed"]))
    for n, m in enumerate(g["meta"]):
        pl = Planes(W, H)
        px, py = int(m[0]), int(m[1])
        for k in range(3):
            win = g["win%d_%d" % (n, k)]; x0, y0 = (int(v) for v in g["pos%d_%d" % (n, k)])
            ya, xa = MARGIN_Y + py + y0, MARGIN_Y + px + x0
            pl.bufY[ya:ya + win.shape[0], xa:xa + win.shape[1]] = win
        pl.bufY[0, 0] = 0
        yield pl, Y, [int(v) for v in m]


def encoder_rdoq_calls():
    """tests/golden/encoder_rdoq_calls.npz (oracle/make_golden8.py): 350 xRateDistOptQuant calls of a real encode, each with the
    context-evolved bit-estimate table the reference's estBit had just written"""
    g = load("encoder_rdoq_calls.npz")
    off = 0
    for hd, lam, tab in zip(g["hd"], g["lam"], g["tab"]):
        log2, comp, intra, scan, tr, qp, bd, sh, as_in, as_out = (int(v) for v in hd)
        n = 1 << (2 * log2)
        yield dict(log2=log2, comp=comp, intra=intra, scan=scan, tr=tr, qp=qp, bd=bd, sh=sh, lam=float(lam), as_in=as_in, asum=as_out,
                   src=np.ascontiguousarray(g["src"][off:off + n], np.int32), eb=np.ascontiguousarray(tab, np.int32),
                   out=np.ascontiguousarray(g["dst"][off:off + n], np.int32))
        off += n


RQT_CFG = np.dtype([("log2_cu", "<i4"), ("qp", "<i4", (3,)), ("bit_depth_y", "<i4"), ("bit_depth_c", "<i4"), ("sign_hide", "<i4"), ("use_ts", "<i4"), ("log2_max_tu", "<i4"),
                    ("log2_min_tu_in_cu", "<i4"), ("inter_split_flag", "<i4"), ("pad", "<i4"), ("lambda_rd", "<f8"), ("lambda_rdoq", "<f8", (3,)), ("dist_weight", "<f8", (3,))])


def encoder_rqt_calls():
    """tests/golden/encoder_rqt_calls.npz (oracle/make_golden9.py): xEstimateResidualQT calls of two real encodes (plain lenslet and the
    sharp-edged frame on which transform skip wins): cfg (hop_o_rqt_cfg image), coder in / out (150 states + fraction), residual planes,
    cost / bits / dist / zero_dist, tr_idx | cbf[3] | tskip[3] (7 x 256), the chosen levels"""
    g = load("encoder_rqt_calls.npz")
    ro = fo = 0
    for i in range(len(g["cost"])):
        cfg = g["cfg"][i]; cu = 1 << int(cfg["log2_cu"]); n = cu * cu * 3 // 2
        yield dict(cfg=cfg, cin=g["cin"][i], cout=g["cout"][i], resi=np.ascontiguousarray(g["resi"][ro:ro + n]), cost=float(g["cost"][i]),
                   bits=int(g["o4"][i][0]), dist=int(g["o4"][i][1]), zero_dist=int(g["o4"][i][2]), arr=g["arr"][i], fin=np.ascontiguousarray(g["fin"][fo:fo + n]),
                   rec=np.ascontiguousarray(g["rec"][ro:ro + n]), org=np.ascontiguousarray(g["org"][ro:ro + n]), d3=[int(v) for v in g["d3"][i]])
        ro += n; fo += n


class _OCoder(ctypes.Structure):
    _fields_ = [("ctx", ctypes.c_uint8 * 150), ("pad", ctypes.c_uint8 * 2), ("frac", ctypes.c_uint64)]


class _OState(ctypes.Structure):
    _fields_ = [("tr_idx", ctypes.c_uint8 * 256), ("cbf", ctypes.c_uint8 * 768), ("tskip", ctypes.c_uint8 * 768), ("coef", ctypes.c_void_p * 12), ("resi", ctypes.c_void_p * 12)]


def oracle_rqt(cfg, ctx150, frac, resi, pred=None, org=None):
    """hop_o_rqt on one CU.  cfg: RQT_CFG record; ctx150: 150 context states; frac: fraction the coder carries; resi: Y | Cb | Cr residual,
    flat int16.  Returns (cost, bits, dist, zero_dist), arrays (7 x 256: tr_idx, cbf[3], tskip[3]), chosen levels, (ctx out, frac out & 32767)."""
    O = oracle()
    c = np.zeros(1, RQT_CFG); c[0] = cfg
    cu = 1 << int(c[0]["log2_cu"]); n2 = cu * cu
    coder = _OCoder(); ctypes.memmove(ctypes.byref(coder), np.ascontiguousarray(ctx150, np.uint8).tobytes() + b"\0\0" + int(frac).to_bytes(8, "little"), 160)
    st = _OState()
    coefs = [[np.zeros(n2 if k == 0 else n2 // 4, np.int32) for k in range(3)] for _ in range(4)]
    resis = [[np.zeros(n2 if k == 0 else n2 // 4, np.int16) for k in range(3)] for _ in range(4)]
    for l in range(4):
        for k in range(3):
            st.coef[3 * l + k] = coefs[l][k].ctypes.data; st.resi[3 * l + k] = resis[l][k].ctypes.data
    resi = np.ascontiguousarray(resi, np.int16)
    ry, rcb, rcr = resi[:n2], resi[n2:n2 + n2 // 4], resi[n2 + n2 // 4:]
    cost = ctypes.c_double(); bits = ctypes.c_uint32(); dist = ctypes.c_uint32(); zd = ctypes.c_uint32()
    O.hop_o_rqt(c.ctypes.data_as(ctypes.c_void_p), ry.ctypes.data_as(ctypes.c_void_p), cu, rcb.ctypes.data_as(ctypes.c_void_p), rcr.ctypes.data_as(ctypes.c_void_p), cu // 2,
                ctypes.byref(coder), ctypes.byref(st), ctypes.byref(cost), ctypes.byref(bits), ctypes.byref(dist), ctypes.byref(zd))
    arr = np.concatenate([np.frombuffer(bytes(st.tr_idx), np.uint8), np.frombuffer(bytes(st.cbf), np.uint8), np.frombuffer(bytes(st.tskip), np.uint8)]).reshape(7, 256).copy()
    fin = np.zeros(n2 * 3 // 2, np.int32)
    O.hop_o_rqt_final_coeffs(c.ctypes.data_as(ctypes.c_void_p), ctypes.byref(st), fin.ctypes.data_as(ctypes.c_void_p))
    out = (cost.value, bits.value, dist.value, zd.value), arr, fin, (np.frombuffer(bytes(coder.ctx), np.uint8).copy(), int(coder.frac) & 32767)
    if pred is None:
        return out
    # the tail of encodeResAndCalcRdInterCU: root-cbf-zero test on the coder as the quadtree left it, reconstruction, final distortions
    P3 = ctypes.c_void_p * 3
    pr = [np.ascontiguousarray(a, np.int16) for a in (pred[:n2], pred[n2:n2 + n2 // 4], pred[n2 + n2 // 4:])]
    og = [np.ascontiguousarray(a, np.int16) for a in (org[:n2], org[n2:n2 + n2 // 4], org[n2 + n2 // 4:])]
    rc = [np.zeros(n2, np.int16), np.zeros(n2 // 4, np.int16), np.zeros(n2 // 4, np.int16)]
    d3 = (ctypes.c_uint32 * 3)(); fin2 = np.zeros(n2 * 3 // 2, np.int32)
    O.hop_o_inter_cu_finish.restype = ctypes.c_int
    O.hop_o_inter_cu_finish.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    root = O.hop_o_inter_cu_finish(c.ctypes.data, ctypes.addressof(st), ctypes.addressof(coder), cost.value, zd.value, P3(*[a.ctypes.data for a in pr]), P3(*[a.ctypes.data for a in og]),
                                   P3(*[a.ctypes.data for a in rc]), d3, fin2.ctypes.data)
    arr2 = np.concatenate([np.frombuffer(bytes(st.tr_idx), np.uint8), np.frombuffer(bytes(st.cbf), np.uint8), np.frombuffer(bytes(st.tskip), np.uint8)]).reshape(7, 256).copy()
    return out + (dict(root=root, rec=np.concatenate(rc), d3=list(d3), arr=arr2, fin=fin2),)


CU_SYN = np.dtype([("part_size", "<i4"), ("n_pu", "<i4"), ("skip_flag", "<i4"), ("skip_ctx", "<i4"), ("amp_acc", "<i4"), ("is_min_cu", "<i4"), ("max_merge_cand", "<i4"),
                   ("pu", [("merge_flag", "<i4"), ("merge_idx", "<i4"), ("mvd", "<i4", (2,)), ("mvp_idx", "<i4"), ("gt_flag", "<i4"), ("gt", "<i4", (8,))], (4,))])


def encoder_cubits_calls():
    """tests/golden/encoder_cubits_calls.npz (oracle/make_golden10.py): xAddSymbolBitsInter calls of two real encodes: cfg, syntax elements (hop_o_cu_syntax
    image), tr_idx | cbf[3] | tskip[3], the CU's levels, coder (150 states + fraction) and the 16 CU-level context states in and out, bits, skip decision"""
    g = load("encoder_cubits_calls.npz")
    o = 0
    for i in range(len(g["bits"])):
        cu = 1 << int(g["cfg"][i]["log2_cu"]); n = cu * cu * 3 // 2
        yield dict(cfg=g["cfg"][i], syn=g["syn"][i], arr=g["arr"][i], coef=np.ascontiguousarray(g["coef"][o:o + n]), cin=g["cin"][i], cuin=g["cuin"][i], cout=g["cout"][i],
                   cuout=g["cuout"][i], bits=int(g["bits"][i]), skipped=int(g["skipped"][i]))
        o += n


INTRA_SYN = np.dtype([("part_nxn", "<i4"), ("skip_flag", "<i4"), ("skip_ctx", "<i4"), ("is_min_cu", "<i4"), ("luma_dir", "<i4", (4,)), ("preds", "<i4", (4, 3)), ("pred_num", "<i4", (4,)),
                      ("chroma_is_dm", "<i4"), ("chroma_dir", "<i4")])


def encoder_intrabits_calls():
    """tests/golden/encoder_intrabits_calls.npz (oracle/make_golden12.py): xGetIntraBitsQT calls of two real encodes: cfg, syntax (hop_o_intra_syntax image), node
    (tr_depth, part, luma, chroma), tr_idx | cbf[3] | tskip[3], levels of the current tree in the CU layout, coder / CU-level contexts in and out, bits"""
    g = load("encoder_intrabits_calls.npz")
    o = 0
    for i in range(len(g["bits"])):
        cu = 1 << int(g["cfg"][i]["log2_cu"]); n = cu * cu * 3 // 2
        yield dict(cfg=g["cfg"][i], syn=g["syn"][i], nd=[int(v) for v in g["nd"][i]], arr=g["arr"][i], coef=np.ascontiguousarray(g["coef"][o:o + n]), cin=g["cin"][i],
                   cuin=g["cuin"][i], cout=g["cout"][i], cuout=g["cuout"][i], bits=int(g["bits"][i]))
        o += n


def encoder_irqt_calls():
    """tests/golden/encoder_irqt_calls.npz (oracle/make_golden14.py): xRecurIntraCodingQT calls (luma tree of an intra PU) of two real encodes: cfg, syntax, nd =
    (tr_depth, part, bCheckFirst, TransformSkipFast, strong smoothing), neighbour flags [341][36], the CU's luma original, the reconstruction picture from (-1, -1)
    of the CU ((2 cu + 1)^2), arrays / coder (160 B) / CU contexts in and out, cost, distortion, the CU's picture block and the chosen luma levels after"""
    g = load("encoder_irqt_calls.npz")
    o1 = o2 = 0
    for i in range(len(g["cost"])):
        cu = 1 << int(g["cfg"][i]["log2_cu"]); n = cu * cu; W = 2 * cu + 1
        yield dict(cfg=g["cfg"][i], syn=g["syn"][i], nd=[int(v) for v in g["nd"][i]], avail=g["avail"][i], org=np.ascontiguousarray(g["org"][o1:o1 + n]),
                   win=np.ascontiguousarray(g["win"][o2:o2 + W * W]), ain=g["ain"][i], cin=g["cin"][i], cuin=g["cuin"][i], cost=float(g["cost"][i]), dist=int(g["dist"][i]),
                   aout=g["aout"][i], cout=g["cout"][i], cuout=g["cuout"][i], rec=np.ascontiguousarray(g["rec"][o1:o1 + n]), fin=np.ascontiguousarray(g["fin"][o1:o1 + n]))
        o1 += n; o2 += W * W


class _OIntraIn(ctypes.Structure):
    _fields_ = [("org", ctypes.c_void_p), ("org_stride", ctypes.c_int), ("rec", ctypes.c_void_p), ("rec_stride", ctypes.c_int), ("avail", ctypes.c_void_p),
                ("strong", ctypes.c_int), ("check_first", ctypes.c_int), ("ts_fast", ctypes.c_int)]


def oracle_intra_rqt(cfg, syn, nd, avail, org, win, arr_in, coder160, cu20):
    """hop_o_intra_rqt on one PU.  win: the reconstruction picture from (-1, -1) of the CU, (2 cu + 1)^2 (written as the search goes).  Returns cost, dist,
    arrays (7 x 256), coder (160 bytes), CU contexts, the window after, the chosen luma levels (CU layout)."""
    O = oracle()
    c = np.zeros(1, RQT_CFG); c[0] = cfg
    y = np.zeros(1, INTRA_SYN); y[0] = syn
    cu = 1 << int(c[0]["log2_cu"]); n2 = cu * cu; W = 2 * cu + 1
    coder = _OCoder(); ctypes.memmove(ctypes.byref(coder), np.ascontiguousarray(coder160).tobytes(), 160)
    cuctx = np.ascontiguousarray(cu20, np.uint8).copy()
    st = _OState()
    a = np.ascontiguousarray(arr_in, np.uint8).reshape(-1)
    ctypes.memmove(ctypes.addressof(st), a.tobytes(), 1792)            # tr_idx, cbf[3], tskip[3] lead the struct
    coefs = [[np.zeros(n2 if k == 0 else n2 // 4, np.int32) for k in range(3)] for _ in range(4)]
    recs = [np.zeros(n2, np.int16) for _ in range(4)]
    for l in range(4):
        for k in range(3):
            st.coef[3 * l + k] = coefs[l][k].ctypes.data
        st.resi[3 * l] = recs[l].ctypes.data
    win = np.ascontiguousarray(win, np.int16).copy(); org = np.ascontiguousarray(org, np.int16); avail = np.ascontiguousarray(avail, np.uint8)
    inp = _OIntraIn(org.ctypes.data, cu, win.ctypes.data + 2 * (W + 1), W, avail.ctypes.data, nd[4], nd[2], nd[3])
    cost = ctypes.c_double(0.0); dist = ctypes.c_uint32(0)
    O.hop_o_intra_rqt.restype = None
    O.hop_o_intra_rqt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    O.hop_o_intra_rqt(c.ctypes.data, y.ctypes.data, ctypes.addressof(inp), nd[0], nd[1], ctypes.addressof(coder), cuctx.ctypes.data, ctypes.addressof(st),
                      ctypes.addressof(cost), ctypes.addressof(dist))
    arr = np.concatenate([np.frombuffer(bytes(st.tr_idx), np.uint8), np.frombuffer(bytes(st.cbf), np.uint8), np.frombuffer(bytes(st.tskip), np.uint8)]).reshape(7, 256).copy()
    fin = np.zeros(n2 * 3 // 2, np.int32)
    O.hop_o_rqt_final_coeffs(c.ctypes.data_as(ctypes.c_void_p), ctypes.byref(st), fin.ctypes.data_as(ctypes.c_void_p))
    return cost.value, dist.value, arr, np.frombuffer(bytes(coder), np.uint8).copy(), cuctx, win, fin[:n2].copy()


def encoder_isearch_calls():
    """tests/golden/encoder_isearch_calls.npz (oracle/make_golden15.py): estIntraPredQT calls (luma intra search of a CU) of two real encodes: cfg, syntax constants, nd =
    (TransformSkipFast, strong smoothing, candidates for the full RD), dirs = left[4] | above[4] directions outside the CU, sqrt(lambda), neighbour flags of the PUs
    [4][68] and of every node [341][36], the CU's luma original, the reconstruction picture from (-1, -1) of the CU, the CI_CURR_BEST coder (160 B) and CU contexts;
    after: best directions, candidates tested, distortion, arrays, the CU's luma levels, its reconstruction plane, its picture block"""
    g = load("encoder_isearch_calls.npz")
    o1 = o2 = 0
    for i in range(len(g["dist"])):
        cu = 1 << int(g["cfg"][i]["log2_cu"]); n = cu * cu; W = 2 * cu + 1
        yield dict(cfg=g["cfg"][i], syn=g["syn"][i], nd=[int(v) for v in g["nd"][i]], dirs=g["dirs"][i], sql=float(g["sql"][i]), rough=g["rough"][i], avail=g["avail"][i],
                   org=np.ascontiguousarray(g["org"][o1:o1 + n]), win=np.ascontiguousarray(g["win"][o2:o2 + W * W]), cin=g["cin"][i], cuin=g["cuin"][i], best=g["best"][i],
                   ncand=g["ncand"][i], dist=int(g["dist"][i]), aout=g["aout"][i], coef=np.ascontiguousarray(g["coef"][o1:o1 + n]),
                   reco=np.ascontiguousarray(g["reco"][o1:o1 + n]), rec=np.ascontiguousarray(g["rec"][o1:o1 + n]))
        o1 += n; o2 += W * W


class _OSearchIn(ctypes.Structure):
    _fields_ = [("left_dir", ctypes.c_int * 4), ("above_dir", ctypes.c_int * 4), ("rough_flags", ctypes.c_void_p), ("sqrt_lambda", ctypes.c_double), ("num_full_rd", ctypes.c_int)]


def oracle_intra_luma_search(cfg, syn, nd, dirs, sqrt_lambda, rough, avail, org, win, coder160, cu20):
    """hop_o_intra_luma_search on one CU.  Returns best directions, candidates tested, distortion, arrays (7 x 256), luma levels, reconstruction plane, the window after."""
    O = oracle()
    c = np.zeros(1, RQT_CFG); c[0] = cfg
    y = np.zeros(1, INTRA_SYN); y[0] = syn
    cu = 1 << int(c[0]["log2_cu"]); n2 = cu * cu; W = 2 * cu + 1
    coder = _OCoder(); ctypes.memmove(ctypes.byref(coder), np.ascontiguousarray(coder160).tobytes(), 160)
    cuctx = np.ascontiguousarray(cu20, np.uint8).copy()
    st = _OState()
    coefs = [[np.zeros(n2 if k == 0 else n2 // 4, np.int32) for k in range(3)] for _ in range(4)]
    recs = [np.zeros(n2, np.int16) for _ in range(4)]
    for l in range(4):
        for k in range(3):
            st.coef[3 * l + k] = coefs[l][k].ctypes.data
        st.resi[3 * l] = recs[l].ctypes.data
    win = np.ascontiguousarray(win, np.int16).copy(); org = np.ascontiguousarray(org, np.int16); avail = np.ascontiguousarray(avail, np.uint8); rough = np.ascontiguousarray(rough, np.uint8)
    inp = _OIntraIn(org.ctypes.data, cu, win.ctypes.data + 2 * (W + 1), W, avail.ctypes.data, nd[1], 0, nd[0])
    sin = _OSearchIn((ctypes.c_int * 4)(*[int(v) for v in dirs[:4]]), (ctypes.c_int * 4)(*[int(v) for v in dirs[4:]]), rough.ctypes.data, sqrt_lambda, nd[2])
    best = (ctypes.c_int * 4)(); ncand = (ctypes.c_int * 4)(); dist = ctypes.c_uint32(0)
    coef_y = np.zeros(n2, np.int32); reco = np.zeros(n2, np.int16)
    O.hop_o_intra_luma_search.restype = None
    O.hop_o_intra_luma_search.argtypes = [ctypes.c_void_p] * 12
    O.hop_o_intra_luma_search(c.ctypes.data, y.ctypes.data, ctypes.addressof(inp), ctypes.addressof(sin), ctypes.addressof(coder), cuctx.ctypes.data, ctypes.addressof(st),
                              ctypes.addressof(best), coef_y.ctypes.data, reco.ctypes.data, ctypes.addressof(dist), ctypes.addressof(ncand))
    arr = np.concatenate([np.frombuffer(bytes(st.tr_idx), np.uint8), np.frombuffer(bytes(st.cbf), np.uint8), np.frombuffer(bytes(st.tskip), np.uint8)]).reshape(7, 256).copy()
    return list(best), list(ncand), dist.value, arr, coef_y, reco, win


def encoder_csearch_calls():
    """tests/golden/encoder_csearch_calls.npz (oracle/make_golden16.py): estIntraPredChromaQT calls (chroma intra search of a CU) of two real encodes: cfg, syntax (the luma
    directions decided before), nd = (TransformSkipFast, -, -, -), neighbour flags of every node [341][36], chroma originals Cb | Cr, the chroma reconstruction pictures
    from (-1, -1) of the CU ((size + 1)^2 each), arrays in, the CI_CURR_BEST coder and CU contexts; after: direction, distortion, arrays, chroma levels Cb | Cr,
    reconstruction planes Cb | Cr, picture blocks Cb | Cr"""
    g = load("encoder_csearch_calls.npz")
    o1 = o2 = 0
    for i in range(len(g["dist"])):
        cu = 1 << int(g["cfg"][i]["log2_cu"]); n = cu * cu // 2; W = cu + 1
        yield dict(cfg=g["cfg"][i], syn=g["syn"][i], nd=[int(v) for v in g["nd"][i]], avail=g["avail"][i], org=np.ascontiguousarray(g["org"][o1:o1 + n]),
                   win=np.ascontiguousarray(g["win"][o2:o2 + 2 * W * W]), ain=g["ain"][i], cin=g["cin"][i], cuin=g["cuin"][i], mode=int(g["mode"][i]), dist=int(g["dist"][i]),
                   aout=g["aout"][i], coef=np.ascontiguousarray(g["coef"][o1:o1 + n]), reco=np.ascontiguousarray(g["reco"][o1:o1 + n]), rec=np.ascontiguousarray(g["rec"][o1:o1 + n]))
        o1 += n; o2 += 2 * W * W


class _OChromaIn(ctypes.Structure):
    _fields_ = [("org_cb", ctypes.c_void_p), ("org_cr", ctypes.c_void_p), ("org_stride", ctypes.c_int), ("rec_cb", ctypes.c_void_p), ("rec_cr", ctypes.c_void_p),
                ("rec_stride", ctypes.c_int), ("avail", ctypes.c_void_p), ("ts_fast", ctypes.c_int)]


def oracle_intra_chroma_search(cfg, syn, ts_fast, avail, org, win, arr_in, coder160, cu20):
    """hop_o_intra_chroma_search on one CU.  org: Cb | Cr originals (half size squared each); win: Cb | Cr reconstruction pictures from (-1, -1), (size + 1)^2 each.
    Returns direction, distortion, arrays (7 x 256), levels Cb | Cr, reconstruction planes Cb | Cr, the windows after."""
    O = oracle()
    c = np.zeros(1, RQT_CFG); c[0] = cfg
    y = np.zeros(1, INTRA_SYN); y[0] = syn
    cu = 1 << int(c[0]["log2_cu"]); n2 = cu * cu; h2 = n2 // 4; half = cu // 2; W = cu + 1
    coder = _OCoder(); ctypes.memmove(ctypes.byref(coder), np.ascontiguousarray(coder160).tobytes(), 160)
    cuctx = np.ascontiguousarray(cu20, np.uint8).copy()
    st = _OState()
    ctypes.memmove(ctypes.addressof(st), np.ascontiguousarray(arr_in, np.uint8).reshape(-1).tobytes(), 1792)
    coefs = [[np.zeros(n2 if k == 0 else h2, np.int32) for k in range(3)] for _ in range(4)]
    recs = [[np.zeros(n2 if k == 0 else h2, np.int16) for k in range(3)] for _ in range(4)]
    for l in range(4):
        for k in range(3):
            st.coef[3 * l + k] = coefs[l][k].ctypes.data; st.resi[3 * l + k] = recs[l][k].ctypes.data
    win = np.ascontiguousarray(win, np.int16).copy(); org = np.ascontiguousarray(org, np.int16); avail = np.ascontiguousarray(avail, np.uint8)
    inp = _OChromaIn(org.ctypes.data, org.ctypes.data + 2 * h2, half, win.ctypes.data + 2 * (W + 1), win.ctypes.data + 2 * W * W + 2 * (W + 1), W, avail.ctypes.data, ts_fast)
    mode = ctypes.c_int(0); dist = ctypes.c_uint32(0)
    coef = np.zeros(2 * h2, np.int32); reco = np.zeros(2 * h2, np.int16)
    O.hop_o_intra_chroma_search.restype = None
    O.hop_o_intra_chroma_search.argtypes = [ctypes.c_void_p] * 12
    O.hop_o_intra_chroma_search(c.ctypes.data, y.ctypes.data, ctypes.addressof(inp), ctypes.addressof(coder), cuctx.ctypes.data, ctypes.addressof(st), ctypes.addressof(mode),
                                ctypes.addressof(dist), coef.ctypes.data, coef.ctypes.data + 4 * h2, reco.ctypes.data, reco.ctypes.data + 2 * h2)
    arr = np.concatenate([np.frombuffer(bytes(st.tr_idx), np.uint8), np.frombuffer(bytes(st.cbf), np.uint8), np.frombuffer(bytes(st.tskip), np.uint8)]).reshape(7, 256).copy()
    return mode.value, dist.value, arr, coef, reco, win


def encoder_intracu_calls():
    """tests/golden/encoder_intracu_calls.npz (oracle/make_golden17.py): the bit counts of finished intra CUs (xCheckRDCostIntra) of two real encodes: cfg, syntax, tr_idx |
    cbf[3] | tskip[3], the CU's levels Y | Cb | Cr, coder (160 B) / CU contexts in and out, bits, distortion"""
    g = load("encoder_intracu_calls.npz")
    o = 0
    for i in range(len(g["bits"])):
        cu = 1 << int(g["cfg"][i]["log2_cu"]); n = cu * cu * 3 // 2
        yield dict(cfg=g["cfg"][i], syn=g["syn"][i], arr=g["arr"][i], coef=np.ascontiguousarray(g["coef"][o:o + n]), cin=g["cin"][i], cuin=g["cuin"][i], cout=g["cout"][i],
                   cuout=g["cuout"][i], bits=int(g["bits"][i]), dist=int(g["dist"][i]))
        o += n


def encoder_cuskip_calls():
    """tests/golden/encoder_cuskip_calls.npz (oracle/make_golden18.py): residual-free candidates (encodeResAndCalcRdInterCU, bSkipRes) of two real encodes: cfg, nd = (skip
    context, merge index, MaxNumMergeCand, -), prediction and original planes Y | Cb | Cr of the CU, coder (160 B) and the 16 CU-level context states in and out, o4 =
    (bits, distortion Y, Cb, Cr), cost"""
    g = load("encoder_cuskip_calls.npz")
    o = 0
    for i in range(len(g["cost"])):
        cu = 1 << int(g["cfg"][i]["log2_cu"]); n = cu * cu * 3 // 2
        yield dict(cfg=g["cfg"][i], nd=[int(v) for v in g["nd"][i]], pred=np.ascontiguousarray(g["pred"][o:o + n]), org=np.ascontiguousarray(g["org"][o:o + n]), cin=g["cin"][i],
                   cuin=g["cuin"][i], cout=g["cout"][i], cuout=g["cuout"][i], o4=[int(v) for v in g["o4"][i]], cost=float(g["cost"][i]))
        o += n
