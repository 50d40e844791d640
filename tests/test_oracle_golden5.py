"""CPU: the restatement's ME chain replayed on the calls of a real encode (tests/golden/encoder_calls.npz, sampled by
oracle/make_golden7.py from the shim encoder whose bitstream equals the reference's): 95 PUs of 20 shapes incl. the AMP shapes,
real predictors / AMVP lists / causality offsets, SS-reference states with sentinel regions.  Also proves that the fixture holds
every sample the members read (anything missing is a sentinel here and would change the result)."""
import ctypes

import numpy as np

from goldutil import encoder_calls
from hoputil import oracle, p16


def test_me_chain_on_encoder_calls():
    O = oracle()
    n = 0
    shapes = set()
    for pl, Y, m in encoder_calls():
        px, py, w, h = m[0:4]
        l, r, t, b, ox, oy, predx, predy, lc, fen, had, namvp = m[4:16]
        amvp = (ctypes.c_int * 4)(*m[16:20])
        org = np.ascontiguousarray(Y[py:py + h, px:px + w])
        out = (ctypes.c_int64 * 32)()
        O.hop_o_me_pu(p16(org), w, pl.ptr00(0), pl.sy, px, py, w, h, l, r, t, b, ox, oy, predx, predy, namvp, amvp, lc, fen, had, 8, 3, out)
        got = list(out)
        assert got[0:3] == m[20:23] and got[3] == 0, (m[:20], got[:4], m[20:23])
        assert got[4:9] == m[23:28], (m[:20], got[4:9], m[23:28])
        want_gt = m[28:44]       # flag, gt[8], cost, mv, half, qter
        assert got[9:25] == want_gt, (m[:20], got[9:25], want_gt)
        shapes.add((w, h)); n += 1
    assert n >= 90 and len(shapes) == 20


def test_rdoq_on_encoder_calls():
    """350 xRateDistOptQuant calls of the same encode (tests/golden/encoder_rdoq_calls.npz), each with the table estBit had written
    from the contexts as they stood: restatement against what the call returned inside the encoder"""
    from goldutil import encoder_rdoq_calls
    O = oracle()
    O.hop_o_rdoq.restype = ctypes.c_int
    O.hop_o_rdoq.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 8 + [ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
    n = 0
    for c in encoder_rdoq_calls():
        d = np.zeros(len(c["src"]), np.int32); a = ctypes.c_uint32(c["as_in"])
        assert O.hop_o_rdoq(c["src"].ctypes.data, d.ctypes.data, c["log2"], c["comp"], c["intra"], c["scan"], c["tr"], c["qp"], c["bd"], c["sh"],
                            c["lam"], c["eb"].ctypes.data, ctypes.byref(a)) == 0
        assert a.value == c["asum"] and np.array_equal(d, c["out"]), (n, c["log2"], c["comp"], c["intra"])
        n += 1
    assert n == 350


def test_rqt_on_encoder_calls():
    """row a8b: the restatement of xEstimateResidualQT (oracle/hop_oracle_rqt.c) on 73 calls recorded inside the encoder: cost, bits,
    distortions, transform depth / cbf / transform-skip arrays, the chosen levels and the coder state it leaves"""
    from goldutil import encoder_rqt_calls, oracle_rqt
    n = ts = zeros = 0
    for c in encoder_rqt_calls():
        pred = (c["org"] - c["resi"]).astype(np.int16)
        res, arr, fin, (cx, fr), tail = oracle_rqt(c["cfg"], c["cin"]["ctx"], int(c["cin"]["frac"]), c["resi"], pred, c["org"])
        # what encodeResAndCalcRdInterCU made of it: root-cbf-zero test, reconstruction, the three final distortions
        assert np.array_equal(tail["rec"], c["rec"]) and tail["d3"] == c["d3"], (n, tail["root"], tail["d3"], c["d3"])
        zeros += int(tail["root"] == 0)
        parts = (1 << (2 * int(c["cfg"]["log2_cu"]))) // 16
        assert res == (c["cost"], c["bits"], c["dist"], c["zero_dist"]), n
        assert np.array_equal(arr[:, :parts], c["arr"].reshape(7, 256)[:, :parts]) and np.array_equal(fin, c["fin"]), n
        assert np.array_equal(cx, c["cout"]["ctx"]) and fr == (int(c["cout"]["frac"]) & 32767), n
        ts += int(arr[4:, :parts].any()); n += 1
    assert n == 73 and ts >= 15 and zeros >= 1, zeros


def test_cu_bits_on_encoder_calls():
    """the CU-level syntax bits of an SS/GT CU (restatement of xAddSymbolBitsInter: skip, merge, partition, MVD, MVP index, GT flag and vectors, root cbf,
    transform tree in bitstream order) on 57 calls recorded inside the encoder: bits, the skip decision and every context state afterwards"""
    from goldutil import encoder_cubits_calls, RQT_CFG, CU_SYN, _OCoder, _OState
    O = oracle()
    O.hop_o_inter_cu_bits.restype = ctypes.c_uint32
    n = 0; shapes = set()
    for c in encoder_cubits_calls():
        cfg = np.zeros(1, RQT_CFG); cfg[0] = c["cfg"]; syn = np.zeros(1, CU_SYN); syn[0] = c["syn"]
        st = _OState(); ctypes.memmove(ctypes.byref(st), c["arr"].tobytes(), 256 * 7)
        coder = _OCoder(); ctypes.memmove(ctypes.byref(coder), c["cin"].tobytes(), 160)
        cu = c["cuin"].copy(); sk = ctypes.c_int(0)
        bits = O.hop_o_inter_cu_bits(cfg.ctypes.data_as(ctypes.c_void_p), syn.ctypes.data_as(ctypes.c_void_p), ctypes.byref(st), c["coef"].ctypes.data_as(ctypes.c_void_p),
                                     ctypes.byref(coder), cu.ctypes.data_as(ctypes.c_void_p), ctypes.byref(sk))
        assert (bits, sk.value) == (c["bits"], c["skipped"]), (n, bits, c["bits"])
        assert bytes(coder.ctx) == c["cout"]["ctx"].tobytes() and int(coder.frac) == int(c["cout"]["frac"]) and np.array_equal(cu, c["cuout"]), n
        shapes.add(int(syn[0]["part_size"])); n += 1
    assert n == 57 and len(shapes) >= 7


def test_intra_mode_bits_on_encoder_calls():
    """xModeBitsIntra: 600 calls recorded inside the encoder (state of the prev_intra_luma_pred_flag context, carried fraction, mode, most probable modes)
    against the restatement; plus the candidate-list logic on a hand-checkable case"""
    from goldutil import load
    O = oracle()
    O.hop_o_intra_mode_bits.restype = ctypes.c_uint32
    rec = load("encoder_modebits_calls.npz")["rec"]
    kinds = set()
    for st, fr, mode, p0, p1, p2, pn, bits in rec:
        s = ctypes.c_uint8(int(st)); f = ctypes.c_uint64(int(fr)); preds = (ctypes.c_int * 3)(int(p0), int(p1), int(p2))
        assert O.hop_o_intra_mode_bits(ctypes.byref(s), ctypes.byref(f), int(mode), preds, int(pn)) == int(bits)
        kinds.add(int(mode) in (int(p0), int(p1), int(p2)))
    assert kinds == {True, False}
    satd = (ctypes.c_uint32 * 35)(*[1000 + 10 * m for m in range(35)]); satd[20] = 5; satd[7] = 6
    modes = (ctypes.c_uint32 * 11)(); costs = (ctypes.c_double * 8)(); preds = (ctypes.c_int * 3)(0, 1, 26)
    O.hop_o_intra_cand_list.restype = ctypes.c_int
    O.hop_o_intra_cand_list.argtypes = [ctypes.c_void_p, ctypes.c_uint8, ctypes.c_uint32, ctypes.c_double, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    n = O.hop_o_intra_cand_list(satd, 60, 100, 7.6, preds, 3, 3, 3, modes, costs)
    # the two cheap modes, then mode 0 (the most probable mode with the shortest code among the rest); the missing most probable modes 1 and 26 appended
    assert n == 5 and list(modes)[:5] == [20, 7, 0, 1, 26] and costs[0] < costs[1] < costs[2]


def test_intra_cu_bits_on_encoder_calls():
    """xGetIntraBitsQT (intra CU header, split / cbf tree, levels from a node downwards): the restatement on 87 calls recorded inside the encoder, fed with the
    levels of the current tree in the CU layout (every layer pointer at the same array): bits and every context state afterwards"""
    from goldutil import encoder_intrabits_calls, RQT_CFG, INTRA_SYN, _OCoder, _OState
    O = oracle()
    O.hop_o_intra_cu_bits.restype = ctypes.c_uint32
    n = 0; kinds = set()
    for c in encoder_intrabits_calls():
        cfg = np.zeros(1, RQT_CFG); cfg[0] = c["cfg"]; syn = np.zeros(1, INTRA_SYN); syn[0] = c["syn"]
        cu2 = 1 << (2 * int(cfg[0]["log2_cu"]))
        st = _OState(); ctypes.memmove(ctypes.byref(st), c["arr"].tobytes(), 256 * 7)
        planes = [np.ascontiguousarray(c["coef"][:cu2]), np.ascontiguousarray(c["coef"][cu2:cu2 + cu2 // 4]), np.ascontiguousarray(c["coef"][cu2 + cu2 // 4:])]
        for l in range(4):
            for k in range(3): st.coef[3 * l + k] = planes[k].ctypes.data
        coder = _OCoder(); ctypes.memmove(ctypes.byref(coder), c["cin"].tobytes(), 160)
        cu = c["cuin"].copy()
        bits = O.hop_o_intra_cu_bits(cfg.ctypes.data_as(ctypes.c_void_p), syn.ctypes.data_as(ctypes.c_void_p), ctypes.byref(st), c["nd"][0], c["nd"][1], c["nd"][2], c["nd"][3],
                                     ctypes.byref(coder), cu.ctypes.data_as(ctypes.c_void_p))
        assert bits == c["bits"], (n, bits, c["bits"])
        assert bytes(coder.ctx) == c["cout"]["ctx"].tobytes() and int(coder.frac) == int(c["cout"]["frac"]) and np.array_equal(cu, c["cuout"]), n
        kinds.add((int(syn[0]["part_nxn"]), c["nd"][2], c["nd"][3])); n += 1
    assert n == 87 and len(kinds) >= 4


def test_intra_rqt_on_encoder_calls():
    """xRecurIntraCodingQT, luma only (the transform tree of an intra PU with its 4x4 transform-skip retry, reconstruction into the picture as it goes, bits via
    xGetIntraBitsQT): the restatement on 47 calls recorded inside the encoder - cost, distortion, depth / cbf / transform-skip arrays, coder and CU-level context
    states, the picture block and the chosen levels afterwards"""
    from goldutil import encoder_irqt_calls, oracle_intra_rqt
    n = 0; kinds = set()
    for c in encoder_irqt_calls():
        cu = 1 << int(c["cfg"]["log2_cu"]); W = 2 * cu + 1
        cost, dist, arr, coder, cuctx, win, fin = oracle_intra_rqt(c["cfg"], c["syn"], c["nd"], c["avail"], c["org"], c["win"], c["ain"], c["cin"].tobytes(), c["cuin"])
        assert cost == c["cost"] and dist == c["dist"], (n, cost, c["cost"], dist, c["dist"])
        assert np.array_equal(arr.reshape(-1)[[*range(256), *range(256, 512), *range(1024, 1280)]], c["aout"][[*range(256), *range(256, 512), *range(1024, 1280)]]), n
        assert coder.tobytes() == c["cout"].tobytes() and np.array_equal(cuctx, c["cuout"]), n
        assert np.array_equal(win.reshape(W, W)[1:1 + cu, 1:1 + cu].reshape(-1), c["rec"]), n
        parts = (cu // 4) ** 2; p0 = c["nd"][1]; np_ = parts >> (2 * c["nd"][0])
        assert np.array_equal(fin[16 * p0:16 * (p0 + np_)], c["fin"][16 * p0:16 * (p0 + np_)]), n
        kinds.add((cu, int(c["syn"]["part_nxn"]), c["nd"][2], int(arr[0, p0:p0 + np_].max()), int(arr[4, p0:p0 + np_].any()))); n += 1
    assert n == 47 and len(kinds) >= 12 and any(k[4] for k in kinds) and any(k[3] >= 2 for k in kinds), kinds


def test_intra_luma_search_on_encoder_calls():
    """estIntraPredQT, luma only (per PU: most probable modes, 35-mode rough search, candidate list, every candidate through the transform tree with bCheckFirst, the
    best one with the full tree, results kept the way xSetIntraResultQT keeps them): the restatement on 43 calls recorded inside the encoder - directions, number of
    candidates, distortion, arrays, the CU's luma levels, its reconstruction plane and what the picture holds afterwards"""
    from goldutil import encoder_isearch_calls, oracle_intra_luma_search
    n = 0; kinds = set()
    for c in encoder_isearch_calls():
        cu = 1 << int(c["cfg"]["log2_cu"]); W = 2 * cu + 1; parts = (cu // 4) ** 2; npu = 4 if c["syn"]["part_nxn"] else 1
        best, ncand, dist, arr, coef, reco, win = oracle_intra_luma_search(c["cfg"], c["syn"], c["nd"], c["dirs"], c["sql"], c["rough"], c["avail"], c["org"], c["win"],
                                                                          c["cin"].tobytes(), c["cuin"])
        assert best[:npu] == [int(v) for v in c["best"][:npu]] and ncand[:npu] == [int(v) for v in c["ncand"][:npu]] and dist == c["dist"], (n, best, c["best"], dist, c["dist"])
        a = c["aout"].reshape(7, 256)
        assert np.array_equal(arr[[0, 1, 4], :parts], a[[0, 1, 4], :parts]), n
        assert np.array_equal(coef, c["coef"]) and np.array_equal(reco, c["reco"]), n
        assert np.array_equal(win.reshape(W, W)[1:1 + cu, 1:1 + cu].reshape(-1), c["rec"]), n
        kinds.add((cu, npu, ncand[0], int(arr[0, :parts].max()), int(arr[4, :parts].any()))); n += 1
    assert n == 43 and len(kinds) >= 12 and any(k[4] for k in kinds) and any(k[1] == 4 for k in kinds), kinds


def test_intra_chroma_search_on_encoder_calls():
    """estIntraPredChromaQT (the five allowed chroma directions through xRecurIntraChromaCodingQT along the luma tree, the transform-skip retry per component, the chroma
    bits of the CU, the best kept): the restatement on 44 calls recorded inside the encoder - direction, distortion, cbf / transform-skip arrays, the CU's chroma
    levels, its reconstruction planes and what the chroma pictures hold afterwards"""
    from goldutil import encoder_csearch_calls, oracle_intra_chroma_search
    n = 0; kinds = set()
    for c in encoder_csearch_calls():
        cu = 1 << int(c["cfg"]["log2_cu"]); W = cu + 1; half = cu // 2; parts = (cu // 4) ** 2
        mode, dist, arr, coef, reco, win = oracle_intra_chroma_search(c["cfg"], c["syn"], c["nd"][0], c["avail"], c["org"], c["win"], c["ain"], c["cin"].tobytes(), c["cuin"])
        assert mode == c["mode"] and dist == c["dist"], (n, mode, c["mode"], dist, c["dist"])
        a = c["aout"].reshape(7, 256)
        assert np.array_equal(arr[[2, 3, 5, 6], :parts], a[[2, 3, 5, 6], :parts]), n
        assert np.array_equal(coef, c["coef"]) and np.array_equal(reco, c["reco"]), n
        w2 = win.reshape(2, W, W)[:, 1:1 + half, 1:1 + half]
        assert np.array_equal(w2.reshape(-1), c["rec"]), n
        kinds.add((cu, mode, int(arr[5:7, :parts].any()), int(arr[0, :parts].max()))); n += 1
    assert n == 44 and len(kinds) >= 15 and any(k[2] for k in kinds) and len(set(k[1] for k in kinds)) >= 4, kinds


def test_intra_cu_total_bits_on_encoder_calls():
    """the counting part of xCheckRDCostIntra (header, luma and chroma directions, xEncodeTransform with the intra rules on the CU's final levels): the restatement on 40
    calls recorded inside the encoder - bits and every context state afterwards (what the encoder stores as CI_TEMP_BEST)"""
    from goldutil import encoder_intracu_calls, RQT_CFG, INTRA_SYN, _OCoder, _OState
    O = oracle()
    O.hop_o_intra_cu_total_bits.restype = ctypes.c_uint32
    O.hop_o_intra_cu_total_bits.argtypes = [ctypes.c_void_p] * 6
    n = 0; kinds = set()
    for c in encoder_intracu_calls():
        cfg = np.zeros(1, RQT_CFG); cfg[0] = c["cfg"]; syn = np.zeros(1, INTRA_SYN); syn[0] = c["syn"]
        st = _OState(); ctypes.memmove(ctypes.byref(st), c["arr"].tobytes(), 1792)
        coder = _OCoder(); ctypes.memmove(ctypes.byref(coder), c["cin"].tobytes(), 160)
        cu = c["cuin"].copy()
        bits = O.hop_o_intra_cu_total_bits(cfg.ctypes.data, syn.ctypes.data, ctypes.addressof(st), c["coef"].ctypes.data, ctypes.addressof(coder), cu.ctypes.data)
        assert bits == c["bits"], (n, bits, c["bits"])
        assert bytes(coder.ctx) == c["cout"]["ctx"].tobytes() and int(coder.frac) == int(c["cout"]["frac"]) and np.array_equal(cu, c["cuout"]), n
        kinds.add((int(cfg[0]["log2_cu"]), int(syn[0]["part_nxn"]), int(syn[0]["chroma_is_dm"]))); n += 1
    assert n == 40 and len(kinds) >= 6


def test_inter_cu_skip_on_encoder_calls():
    """encodeResAndCalcRdInterCU without residual (bSkipRes): the restatement on the calls recorded inside the encoder - the three distortions of the prediction, bits of skip
    flag + merge index, cost, context states afterwards"""
    from goldutil import encoder_cuskip_calls, RQT_CFG, _OCoder
    O = oracle()
    O.hop_o_inter_cu_skip.restype = ctypes.c_uint32
    O.hop_o_inter_cu_skip.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 6
    P3 = ctypes.c_void_p * 3
    n = 0; kinds = set()
    for c in encoder_cuskip_calls():
        cfg = np.zeros(1, RQT_CFG); cfg[0] = c["cfg"]; cu2 = 1 << (2 * int(cfg[0]["log2_cu"]))
        pl = lambda a: [np.ascontiguousarray(a[:cu2]), np.ascontiguousarray(a[cu2:cu2 + cu2 // 4]), np.ascontiguousarray(a[cu2 + cu2 // 4:])]
        pr, og = pl(c["pred"]), pl(c["org"])
        coder = _OCoder(); ctypes.memmove(ctypes.byref(coder), c["cin"].tobytes(), 160)
        cu = c["cuin"].copy(); d3 = (ctypes.c_uint32 * 3)(); cost = ctypes.c_double()
        bits = O.hop_o_inter_cu_skip(cfg.ctypes.data, c["nd"][0], c["nd"][1], c["nd"][2], P3(*[a.ctypes.data for a in pr]), P3(*[a.ctypes.data for a in og]), ctypes.addressof(coder),
                                     cu.ctypes.data, ctypes.addressof(d3), ctypes.addressof(cost))
        assert [bits] + list(d3) == c["o4"] and cost.value == c["cost"], (n, bits, list(d3), c["o4"])
        assert bytes(coder.ctx) == c["cout"]["ctx"].tobytes() and int(coder.frac) == int(c["cout"]["frac"]) and np.array_equal(cu, c["cuout"]), n
        kinds.add((int(cfg[0]["log2_cu"]), c["nd"][1])); n += 1
    assert n >= 10 and len(kinds) >= 4, (n, kinds)
