"""GPU: rows a9/a10 (TU round trip) and a7 (intra rough search) through the C ABI against the reference-generated
golden vectors and the oracle.  Bit-exact."""
import ctypes
import importlib.util
import os

import numpy as np
from scipy.ndimage import gaussian_filter
import pytest

from goldutil import load
from hoputil import ROOT, lenslet, oracle, p16

pytestmark = pytest.mark.gpu
VP = ctypes.c_void_p


@pytest.fixture(scope="module", params=["leaf-fused", "leaf-staged"])
def hp(request):
    """every test of this file twice: with the transform-unit leaf step as one kernel per batch (a workgroup per TU, the default for the launch-bound batches of the RD
    search) and as the staged pipeline (one lane per TU, the form for batches of thousands) -- hop_set_fused_leaf / HOP_FUSED_LEAF, read when a context is created"""
    spec = importlib.util.spec_from_file_location("hophip", os.path.join(ROOT, "hevc-hop_amd", "hophip.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    old = os.environ.get("HOP_FUSED_LEAF"), os.environ.get("HOP_WALK")
    if request.param == "leaf-staged":
        os.environ["HOP_FUSED_LEAF"] = "0"; os.environ["HOP_WALK"] = "0"     # ... and the candidate chains as batch steps instead of one walk kernel per candidate (k_walk.inl)
    else:
        os.environ.pop("HOP_FUSED_LEAF", None); os.environ.pop("HOP_WALK", None)
    yield m
    for k, v in zip(("HOP_FUSED_LEAF", "HOP_WALK"), old):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def test_intra_rough_golden(hp):
    g = load("intra.npz")
    Y = g["Y"].astype(np.int16); rec = g["rec"].astype(np.int16)
    H, W = Y.shape
    ctx = hp.Context(W, H)
    z = np.zeros((H // 2, W // 2), np.int16)
    ctx.upload_orig(Y, z, z)
    ctx.plane_upload("recon", 0, rec)
    jobs = []
    for (x, y, N, strong), fl in zip(g["jobs"], g["flags"]):
        j = hp.IntraJob(int(x), int(y), int(N), int(strong))
        for k in range(68):
            j.flags[k] = int(fl[k])
        jobs.append(j)
    got = ctx.intra_rough(jobs)
    assert np.array_equal(got, g["satd"])
    ctx.close()


@pytest.mark.parametrize("bd", [8, 10])
def test_tu_roundtrip_vs_oracle(hp, bd):
    """every TU size, DCT/DST/transform-skip, luma and chroma, several QPs, ISS (85) and I-slice (171) rounding"""
    O = oracle()
    O.hop_o_tu_roundtrip.restype = ctypes.c_uint32
    W, H = 128, 128
    Y, Cb, Cr = lenslet(W, H, 15, 12, bitdepth=bd)
    rng = np.random.default_rng(bd)
    P = [np.clip(a + rng.integers(-30, 31, a.shape) * (1 << (bd - 8)), 0, (1 << bd) - 1).astype(np.int16) for a in (Y, Cb, Cr)]
    ctx = hp.Context(W, H, bit_depth=bd)
    ctx.upload_orig(Y, Cb, Cr)
    for c in range(3):
        ctx.plane_upload("pred", c, P[c])
    jobs = []
    for log2 in (2, 3, 4, 5):
        N = 1 << log2
        for comp in (0, 1, 2):
            for k in range(4):
                lim = (W if comp == 0 else W // 2) - N
                xs, ys = int(rng.integers(0, lim // 4 + 1)) * 4, int(rng.integers(0, lim // 4 + 1)) * 4
                x, y = (xs, ys) if comp == 0 else (2 * xs, 2 * ys)
                qp = int(rng.choice([10, 22, 27, 32, 37, 45])) + 6 * (bd - 8)
                jobs.append(hp.TuJob(x, y, comp, log2, int(log2 == 2 and comp == 0 and k % 2 == 0), int(log2 == 2 and k == 3), qp, k % 2))
    # non-overlapping is not required for the results (each job reads org/pred only), but recon is compared per job, so run one by one
    for j in jobs:
        res, lv = ctx.tu_roundtrip([j])
        N = 1 << j.log2_size
        o_, p_ = (Y, Cb, Cr)[j.comp], P[j.comp]
        x, y = (j.x, j.y) if j.comp == 0 else (j.x // 2, j.y // 2)
        org = np.ascontiguousarray(o_[y:y + N, x:x + N]); prd = np.ascontiguousarray(p_[y:y + N, x:x + N])
        lvo = np.zeros(N * N, np.int32); rec = np.zeros((N, N), np.int16); sse = ctypes.c_uint32()
        s = O.hop_o_tu_roundtrip(bd, j.qp_scaled, j.is_i_slice, j.use_dst, j.transform_skip, N, p16(org), p16(prd),
                                 lvo.ctypes.data_as(VP), p16(rec), ctypes.byref(sse))
        assert res[0] == (s, sse.value), (j.comp, N, res, s, sse.value)
        assert np.array_equal(lv, lvo)
        got = ctx.recon_download(j.comp)[y:y + N, x:x + N]
        assert np.array_equal(got, rec)
    ctx.close()


def test_tu_roundtrip_sign_bit_hiding_vs_oracle(hp):
    """row a10 with the PPS's sign_data_hiding flag (TComTrQuant::signBitHidingHDQ after the flat quantiser): hop_tu_roundtrip with sign_hide against the restatement,
    which tests/test_oracle_golden2.py pins to the reference's own xQuant on 600 golden blocks: every TU size, the three scans, DCT / DST / transform skip, both
    rounding offsets; the hiding must have changed levels in a good part of the blocks"""
    O = oracle()
    O.hop_o_tu_roundtrip_sbh.restype = ctypes.c_uint32; O.hop_o_tu_roundtrip.restype = ctypes.c_uint32
    W, H, bd = 128, 128, 8
    Y, Cb, Cr = lenslet(W, H, 15, 44)
    rng = np.random.default_rng(77)
    P = [np.clip(a + rng.integers(-40, 41, a.shape), 0, 255).astype(np.int16) for a in (Y, Cb, Cr)]
    ctx = hp.Context(W, H)
    ctx.upload_orig(Y, Cb, Cr)
    for c in range(3):
        ctx.plane_upload("pred", c, P[c])
    changed = 0; total = 0
    for log2 in (2, 3, 4, 5):
        N = 1 << log2
        for comp in (0, 1, 2):
            for k in range(8):
                lim = (W if comp == 0 else W // 2) - N
                xs, ys = int(rng.integers(0, lim // 4 + 1)) * 4, int(rng.integers(0, lim // 4 + 1)) * 4
                x, y = (xs, ys) if comp == 0 else (2 * xs, 2 * ys)
                qp = int(rng.choice([10, 17, 22, 27, 32]))
                scan = int(rng.integers(0, 3)) if log2 <= 3 else 0
                j = hp.TuJob(x, y, comp, log2, int(log2 == 2 and comp == 0 and k % 2 == 0), int(log2 == 2 and k == 3), qp, k % 2, 1, scan)
                res, lv = ctx.tu_roundtrip([j])
                o_, p_ = (Y, Cb, Cr)[comp], P[comp]
                org = np.ascontiguousarray(o_[ys:ys + N, xs:xs + N]); prd = np.ascontiguousarray(p_[ys:ys + N, xs:xs + N])
                lvo = np.zeros(N * N, np.int32); lv0 = np.zeros(N * N, np.int32); rec = np.zeros((N, N), np.int16); sse = ctypes.c_uint32(); s0 = ctypes.c_uint32()
                s = O.hop_o_tu_roundtrip_sbh(bd, qp, k % 2, j.use_dst, j.transform_skip, N, 1, scan, p16(org), p16(prd), lvo.ctypes.data_as(VP), p16(rec), ctypes.byref(sse))
                assert res[0] == (s, sse.value) and np.array_equal(lv, lvo), (comp, N, scan, res, s, sse.value)
                assert np.array_equal(ctx.recon_download(comp)[ys:ys + N, xs:xs + N], rec)
                O.hop_o_tu_roundtrip(bd, qp, k % 2, j.use_dst, j.transform_skip, N, p16(org), p16(prd), lv0.ctypes.data_as(VP), p16(np.zeros((N, N), np.int16)), ctypes.byref(s0))
                changed += int(not np.array_equal(lv0, lvo)); total += 1
    assert changed > total // 4, (changed, total)
    ctx.close()


def test_tu_golden_transform_identity(hp):
    """reference-generated forward-transform vectors through the kernel: with prediction = 0 and a QP whose flat
    quantiser is the identity on the tested range the levels expose the forward transform output exactly.
    qp_scaled = 4 (scale 16384 = 2^14): level = (|c| * 2^14 + add) >> (14 + tshift) -> compare via the oracle chain."""
    O = oracle()
    O.hop_o_quant_flat.restype = ctypes.c_uint32
    g = load("tq.npz")
    W = H = 64
    ctx = hp.Context(W, H)
    off = 0
    n_checked = 0
    for (N, bd, dst) in g["meta"]:
        N, bd, dst = int(N), int(bd), int(dst)
        n2 = N * N
        blk = g["blocks"][off:off + n2].reshape(N, N); want_f = g["fwd"][off:off + n2]
        off += n2
        if bd != 8 or np.abs(blk).max() > 255:
            continue
        # org - pred = blk with org, pred in [0,255]
        org = np.zeros((H, W), np.int16); prd = np.zeros((H, W), np.int16)
        org[:N, :N] = np.maximum(blk, 0); prd[:N, :N] = np.maximum(-blk, 0)
        z = np.zeros((H // 2, W // 2), np.int16)
        ctx.upload_orig(org, z, z)
        ctx.plane_upload("pred", 0, prd)
        res, lv = ctx.tu_roundtrip([hp.TuJob(0, 0, 0, int(np.log2(N)), dst, 0, 4, 0)])
        want = np.zeros(n2, np.int32)
        c32 = np.ascontiguousarray(want_f.astype(np.int32))
        O.hop_o_quant_flat(8, 4, 0, c32.ctypes.data_as(VP), want.ctypes.data_as(VP), N)
        assert np.array_equal(lv, want), (N, dst)
        n_checked += 1
    assert n_checked >= 20
    ctx.close()


def test_rdoq_golden(hp):
    """row a11: k_rdoq against the golden vectors made by the reference's own xRateDistOptQuant (360 TUs, every size /
    component / scan / depth / bit depth, sign-bit hiding on and off), all TUs in ONE batch with one table per TU"""
    from test_oracle_golden3 import rdoq_cases
    cases = list(rdoq_cases())
    jobs = np.zeros(len(cases), hp.RDOQ_JOB_DTYPE)
    tables = np.stack([c["eb"] for c in cases])
    off = 0
    for i, c in enumerate(cases):
        j = jobs[i]
        j["log2_size"], j["comp"], j["is_intra"], j["scan_idx"], j["tr_depth"] = c["log2"], c["comp"], c["intra"], c["scan"], c["tr"]
        j["qp_scaled"], j["bit_depth"], j["sign_hide"], j["lambda"], j["coeff_offset"], j["estbits_index"] = c["qp"], c["bd"], c["sh"], c["lam"], off, i
        off += len(c["src"])
    src = np.concatenate([c["src"] for c in cases])
    ctx = hp.Context(64, 64)
    dst, asum = ctx.rdoq(jobs, tables, src)
    off = 0
    for i, c in enumerate(cases):
        n = len(c["src"])
        assert int(asum[i]) == c["asum"] and np.array_equal(dst[off:off + n], c["out"]), (i, c["log2"], c["comp"], c["scan"], c["qp"], c["sh"])
        off += n
    # argument checking: chroma 32x32 does not exist, offsets must stay inside the coefficient array
    bad = jobs[:1].copy(); bad["log2_size"], bad["comp"] = 5, 1
    with pytest.raises(hp.HopError):
        ctx.rdoq(bad, tables, src)
    bad = jobs[:1].copy(); bad["coeff_offset"] = len(src) - 3
    with pytest.raises(hp.HopError):
        ctx.rdoq(bad, tables, src)
    ctx.close()


def test_rdoq_encoder_calls(hp):
    """hop_rdoq on 350 xRateDistOptQuant calls sampled from a real encode (tests/golden/encoder_rdoq_calls.npz, oracle/make_golden8.py):
    real transform coefficients, the context-evolved table of each call, intra scans, every size; one batch"""
    from goldutil import encoder_rdoq_calls
    cases = list(encoder_rdoq_calls())
    jobs = np.zeros(len(cases), hp.RDOQ_JOB_DTYPE)
    off = 0
    for i, c in enumerate(cases):
        j = jobs[i]
        j["log2_size"], j["comp"], j["is_intra"], j["scan_idx"], j["tr_depth"] = c["log2"], c["comp"], c["intra"], c["scan"], c["tr"]
        j["qp_scaled"], j["bit_depth"], j["sign_hide"], j["lambda"], j["coeff_offset"], j["estbits_index"] = c["qp"], c["bd"], c["sh"], c["lam"], off, i
        off += len(c["src"])
    ctx = hp.Context(64, 64)
    dst, asum = ctx.rdoq(jobs, np.stack([c["eb"] for c in cases]), np.concatenate([c["src"] for c in cases]))
    off = 0
    for i, c in enumerate(cases):
        n = len(c["src"])
        assert int(asum[i]) == c["asum"] and np.array_equal(dst[off:off + n], c["out"]), (i, c["log2"], c["comp"], c["intra"])
        off += n
    ctx.close()


def test_coeff_bits_golden(hp):
    """CABAC bit estimator: k_coeff_bits (one lane per TU) against the reference-generated chains: every TU of a chain starts
    from the context states the previous one left (fed back through ctx_out), bits and final states must match"""
    g = load("cabac.npz")
    par, coef, bits, chains, finals, init = g["par"], g["coef"], g["bits"], g["chains"], g["finals"], g["init"]
    ctx = hp.Context(64, 64)
    # step k of every chain in one batch: the chains are independent, the TUs of a chain are not
    state = np.zeros((len(chains), hp.CABAC_CTX_BYTES), np.uint8)
    for i, (sl, qp, t0, t1) in enumerate(chains):
        state[i, :150] = init[sl, qp]
    step = 0
    while True:
        act = [i for i, (sl, qp, t0, t1) in enumerate(chains) if t0 + step < t1]
        if not act:
            break
        jobs = np.zeros(len(act), hp.COEFF_BITS_JOB_DTYPE)
        for k, i in enumerate(act):
            log2, comp, scan, sh, uts, tsf, off = (int(v) for v in par[chains[i][2] + step])
            j = jobs[k]
            j["log2_size"], j["comp"], j["scan_idx"], j["sign_hide"], j["use_ts"], j["ts_flag"], j["ctx_index"], j["coeff_offset"] = log2, comp, scan, sh, uts, tsf, i, off
        b, out = ctx.coeff_bits(jobs, state, coef)
        for k, i in enumerate(act):
            assert int(b[k]) == int(bits[chains[i][2] + step]), (i, step)
            state[i] = out[k]
        step += 1
    assert step >= 4
    assert np.array_equal(state[:, :150], finals)
    bad = np.zeros(1, hp.COEFF_BITS_JOB_DTYPE); bad["log2_size"], bad["ctx_index"] = 3, len(chains)
    with pytest.raises(hp.HopError):
        ctx.coeff_bits(bad, state, coef)
    ctx.close()


def test_tu_rd_golden(hp):
    """row a8b leaf step through hop_tu_rd (forward transform, estBit, RDOQ, counted bits, inverse path, cbf-zero decision as a
    pipeline of kernels) against the vectors assembled from the reference's own members: 192 TUs laid out in one picture,
    each with its own context snapshot"""
    from test_oracle_golden4 import tu_rd_cases
    cases = list(tu_rd_cases())
    W = H = 256
    org = [np.full((H, W), 128, np.int16), np.full((H // 2, W // 2), 128, np.int16), np.full((H // 2, W // 2), 128, np.int16)]
    jobs = np.zeros(len(cases), hp.TU_RD_JOB_DTYPE)
    for i, c in enumerate(cases):
        N = 1 << c["log2"]; k = c["slot"]
        if c["comp"] == 0: px, py = 32 * (k % 8), 32 * (k // 8)
        else: px, py = 16 * (k % 8), 16 * (k // 8)
        org[c["comp"]][py:py + N, px:px + N] = 128 + c["resi"].reshape(N, N)
        j = jobs[i]
        j["x"], j["y"] = (px, py) if c["comp"] == 0 else (2 * px, 2 * py)
        j["comp"], j["log2_size"], j["qp_scaled"], j["tr_depth"], j["ctx_index"], j["sign_hide"], j["use_ts"], j["bit_depth"] = c["comp"], c["log2"], c["qp"], c["trd"], i, c["sh"], c["uts"], 8
        j["lambda_rdoq"], j["lambda_rd"], j["dist_weight"] = c["lamq"], c["lam"], c["w"]
    ctx = hp.Context(W, H)
    ctx.upload_orig(*org)
    for comp in range(3):
        ctx.plane_upload("pred", comp, np.full(org[comp].shape, 128, np.int16))
    res, lv = ctx.tu_rd(jobs, np.stack([c["st"] for c in cases]))
    off = 0
    for i, c in enumerate(cases):
        n = len(c["resi"])
        got = [int(res[i][k]) for k in ("abs_sum", "cbf", "dist", "zero_dist", "nonzero_dist", "bits", "null_bits")]
        assert got == [int(v) for v in c["out"][:7]], (i, c["log2"], c["comp"], got, c["out"])
        assert float(res[i]["cost"]) == c["cost"] and np.array_equal(lv[off:off + n], c["levels"]), (i, c["log2"], c["comp"])
        off += n
    bad = jobs[:1].copy(); bad["bit_depth"] = 10
    with pytest.raises(hp.HopError):
        ctx.tu_rd(bad, np.stack([c["st"] for c in cases]))
    ctx.close()


def test_tu_intra_golden(hp):
    """row a8 leaf step (xIntraCodingLumaBlk / ChromaBlk after the prediction) through hop_tu_rd with is_intra: levels, bits, distortion,
    cost and the reconstruction written into the reconstruction picture, against the reference-assembled goldens (192 TUs)"""
    g = load("tu_intra.npz")
    W = H = 256
    ctx = hp.Context(W, H)
    ctx.upload_orig(g["orgY"].astype(np.int16), g["orgCb"].astype(np.int16), g["orgCr"].astype(np.int16))
    for comp, k in enumerate(("prdY", "prdCb", "prdCr")):
        ctx.plane_upload("pred", comp, g[k].astype(np.int16))
        ctx.plane_upload("recon", comp, np.zeros(g[k].shape, np.int16))
    par = g["par"]
    jobs = np.zeros(len(par), hp.TU_RD_JOB_DTYPE)
    for i in range(len(par)):
        log2, comp, qp, trd, sh, uts, scan, px, py, off = (int(v) for v in par[i])
        j = jobs[i]
        j["x"], j["y"] = (px, py) if comp == 0 else (2 * px, 2 * py)
        j["comp"], j["log2_size"], j["qp_scaled"], j["tr_depth"], j["ctx_index"], j["sign_hide"], j["use_ts"], j["bit_depth"] = comp, log2, qp, trd, i, sh, uts, 8
        j["is_intra"], j["scan_idx"], j["use_dst"] = 1, scan, 1
        j["lambda_rdoq"], j["lambda_rd"], j["dist_weight"] = (float(v) for v in g["lam"][i])
    res, lv = ctx.tu_rd(jobs, g["st"])
    for i in range(len(par)):
        log2, off = int(par[i][0]), int(par[i][9])
        got = [int(res[i][k]) for k in ("abs_sum", "cbf", "dist", "bits")]
        assert got == [int(g["out"][i][k]) for k in (0, 1, 2, 5)], (i, par[i], got, g["out"][i])
        assert float(res[i]["cost"]) == float(g["cost"][i]) and np.array_equal(lv[off:off + (1 << (2 * log2))], g["levels"][off:off + (1 << (2 * log2))]), (i, par[i])
    for comp, k in enumerate(("recY", "recCb", "recCr")):
        assert np.array_equal(ctx.recon_download(comp), g[k]), k
    ctx.close()


def test_intra_pred_vs_oracle(hp):
    """hop_intra_pred: the prediction of one mode per block (xIntraCodingLumaBlk's first step) against the oracle's predictor -- the same
    code whose SATDs are pinned to the reference for all 35 modes -- on the golden blocks (all sizes, partial availability, strong smoothing)"""
    O = oracle()
    g = load("intra.npz")
    Y = g["Y"].astype(np.int16); rec = g["rec"].astype(np.int16)
    H, W = Y.shape
    ctx = hp.Context(W, H)
    z = np.zeros((H // 2, W // 2), np.int16)
    ctx.upload_orig(Y, z, z)
    ctx.plane_upload("recon", 0, rec)
    rng = np.random.default_rng(8)
    IP = ctypes.POINTER(ctypes.c_int)
    for rnd in range(3):
        modes = rng.integers(0, 35, len(g["jobs"])).astype(np.int32)
        if rnd == 0: modes[:] = np.arange(len(modes)) % 35
        # blocks of one call must not overlap in the prediction picture: one job per call is the simplest way to guarantee it here
        for (x, y, N, strong), fl, mode in zip(g["jobs"], g["flags"], modes):
            x, y, N, strong, mode = int(x), int(y), int(N), int(strong), int(mode)
            j = hp.IntraJob(x, y, N, strong)
            for k in range(68):
                j.flags[k] = int(fl[k])
            ctx.intra_pred([j], [mode])
            got = ctx.pred_download(0)[y:y + N, x:x + N]
            L = (ctypes.c_int * (4 * 64 + 8))(); F = (ctypes.c_int * (4 * 64 + 8))()
            fla = np.ascontiguousarray(fl, np.uint8)
            O.hop_o_intra_fill_refs(p16(rec), W, x, y, N, fla.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), 8, L)
            O.hop_o_intra_smooth(L, N, 8, strong, F)
            want = np.zeros((N, N), np.int16)
            O.hop_o_intra_pred(L, F, N, mode, 8, p16(want))
            assert np.array_equal(got, want), (x, y, N, mode)
        if rnd == 0 and len(g["jobs"]) > 60: break
    ctx.close()


def test_rqt_encoder_calls(hp):
    """row a8b through hop_rqt: the whole residual-quadtree search (transform sizes, transform-skip retry, context chaining, subtree recount,
    split decision) on 73 xEstimateResidualQT calls recorded inside the encoder (tests/golden/encoder_rqt_calls.npz: plain lenslet + the
    sharp-edged frame on which transform skip wins), every CU size, all in one call: cost, bits, distortions, the transform depth / cbf /
    transform-skip arrays, the chosen levels and the coder state left behind"""
    from goldutil import encoder_rqt_calls
    cases = list(encoder_rqt_calls())
    W, H = 512, 64 * ((len(cases) + 7) // 8)
    org = [np.full((H, W), 128, np.int16), np.full((H // 2, W // 2), 128, np.int16), np.full((H // 2, W // 2), 128, np.int16)]
    jobs = np.zeros(len(cases), hp.RQT_JOB_DTYPE)
    snaps = np.zeros((len(cases), hp.CABAC_CTX_BYTES), np.uint8)
    for i, c in enumerate(cases):
        cfg = c["cfg"]; cu = 1 << int(cfg["log2_cu"]); n2 = cu * cu
        x, y = 64 * (i % 8), 64 * (i // 8)
        org[0][y:y + cu, x:x + cu] += c["resi"][:n2].reshape(cu, cu)
        org[1][y // 2:(y + cu) // 2, x // 2:(x + cu) // 2] += c["resi"][n2:n2 + n2 // 4].reshape(cu // 2, cu // 2)
        org[2][y // 2:(y + cu) // 2, x // 2:(x + cu) // 2] += c["resi"][n2 + n2 // 4:].reshape(cu // 2, cu // 2)
        j = jobs[i]
        j["x"], j["y"], j["log2_cu"], j["qp_scaled"], j["ctx_index"] = x, y, int(cfg["log2_cu"]), cfg["qp"], i
        j["sign_hide"], j["use_ts"], j["log2_max_tu"], j["log2_min_tu_in_cu"], j["inter_split_flag"] = cfg["sign_hide"], cfg["use_ts"], cfg["log2_max_tu"], cfg["log2_min_tu_in_cu"], cfg["inter_split_flag"]
        j["lambda_rd"], j["lambda_rdoq"], j["dist_weight"] = cfg["lambda_rd"], cfg["lambda_rdoq"], cfg["dist_weight"][1:]
        snaps[i, :150] = c["cin"]["ctx"]
        left = int(c["cin"]["frac"]) & 32767
        snaps[i, 150], snaps[i, 151] = left & 255, left >> 8
    ctx = hp.Context(W, H)
    ctx.upload_orig(*org)
    for comp in range(3):
        ctx.plane_upload("pred", comp, np.full(org[comp].shape, 128, np.int16))
    res, co, cx = ctx.rqt(jobs, snaps)
    off = 0; ts = 0
    for i, c in enumerate(cases):
        cu = 1 << int(c["cfg"]["log2_cu"]); n = cu * cu * 3 // 2; parts = cu * cu // 16
        r = res[i]
        assert (float(r["cost"]), int(r["bits"]), int(r["dist"]), int(r["zero_dist"])) == (c["cost"], c["bits"], c["dist"], c["zero_dist"]), (i, cu, r["cost"], r["bits"], r["dist"], c["cost"], c["bits"], c["dist"])
        got = np.concatenate([r["tr_idx"][None, :], r["cbf"], r["tskip"]])
        assert np.array_equal(got[:, :parts], c["arr"].reshape(7, 256)[:, :parts]), (i, cu)
        assert np.array_equal(co[off:off + n], c["fin"]), (i, cu)
        assert np.array_equal(cx[i, :150], c["cout"]["ctx"]) and (int(cx[i, 150]) | (int(cx[i, 151]) << 8)) == (int(c["cout"]["frac"]) & 32767), (i, cu)
        ts += int(got[4:, :parts].any()); off += n
    assert ts >= 15
    bad = jobs[:1].copy(); bad["x"] = 4
    with pytest.raises(hp.HopError):
        ctx.rqt(bad, snaps)
    ctx.close()
    # the tail of encodeResAndCalcRdInterCU (hop_rqt_finish) on the same calls, now with the encoder's own prediction and original: the
    # root-cbf-zero test, the reconstruction and the three final distortions the encoder computed (:6807-6810)
    org = [np.zeros((H, W), np.int16), np.zeros((H // 2, W // 2), np.int16), np.zeros((H // 2, W // 2), np.int16)]
    prd = [np.zeros((H, W), np.int16), np.zeros((H // 2, W // 2), np.int16), np.zeros((H // 2, W // 2), np.int16)]
    for i, c in enumerate(cases):
        cu = 1 << int(c["cfg"]["log2_cu"]); n2 = cu * cu
        x, y = 64 * (i % 8), 64 * (i // 8)
        o, p = c["org"], (c["org"] - c["resi"]).astype(np.int16)
        for k, (a, b, w) in enumerate(((0, n2, cu), (n2, n2 + n2 // 4, cu // 2), (n2 + n2 // 4, n2 * 3 // 2, cu // 2))):
            xx, yy = (x, y) if k == 0 else (x // 2, y // 2)
            org[k][yy:yy + w, xx:xx + w] = o[a:b].reshape(w, w); prd[k][yy:yy + w, xx:xx + w] = p[a:b].reshape(w, w)
    ctx = hp.Context(W, H)
    ctx.upload_orig(*org)
    for comp in range(3):
        ctx.plane_upload("pred", comp, prd[comp]); ctx.plane_upload("recon", comp, np.zeros(org[comp].shape, np.int16))
    res2, co2, cx2 = ctx.rqt(jobs, snaps)
    assert np.array_equal(res2["bits"], res["bits"]) and np.array_equal(co2, co)           # the residual is the same, so is the quadtree
    res3, co3, fin = ctx.rqt_finish(jobs, res2, co2, cx2)
    rec = [ctx.recon_download(k) for k in range(3)]
    off = 0; zeros = 0
    for i, c in enumerate(cases):
        cu = 1 << int(c["cfg"]["log2_cu"]); n2 = cu * cu; parts = n2 // 16
        x, y = 64 * (i % 8), 64 * (i // 8)
        assert [int(v) for v in fin[i, 1:]] == c["d3"], (i, cu, fin[i], c["d3"])
        got = np.concatenate([rec[0][y:y + cu, x:x + cu].ravel(), rec[1][y // 2:(y + cu) // 2, x // 2:(x + cu) // 2].ravel(), rec[2][y // 2:(y + cu) // 2, x // 2:(x + cu) // 2].ravel()])
        assert np.array_equal(got, c["rec"]), (i, cu, int(fin[i, 0]))
        if fin[i, 0] == 0:
            zeros += 1
            assert not res3[i]["cbf"][:, :parts].any() and not res3[i]["tr_idx"][:parts].any() and not co3[off:off + n2 * 3 // 2].any()
        else:
            assert np.array_equal(co3[off:off + n2 * 3 // 2], c["fin"]) and np.array_equal(res3[i]["cbf"], res2[i]["cbf"])
        off += n2 * 3 // 2
    assert zeros >= 3
    ctx.close()


def test_rqt_random_vs_oracle(hp):
    """hop_rqt against the restatement (pinned inside the reference encoder) where the recorded calls do not go: 8 and 10 bit, QP 20..42,
    sign hiding and transform skip on and off, the inter_split_flag tree, every CU size, smooth and spiky residuals, context states of all
    five slice types with a non-zero carried fraction"""
    import ctypes
    from goldutil import oracle_rqt, RQT_CFG
    for bd in (8, 10):
        rng = np.random.default_rng(40 + bd)
        W, H = 512, 256
        org = [np.full((H, W), 1 << (bd - 1), np.int16), np.full((H // 2, W // 2), 1 << (bd - 1), np.int16), np.full((H // 2, W // 2), 1 << (bd - 1), np.int16)]
        n = 32
        jobs = np.zeros(n, hp.RQT_JOB_DTYPE); cfgs = np.zeros(n, RQT_CFG); snaps = np.zeros((n, hp.CABAC_CTX_BYTES), np.uint8); resis = []
        ctx = hp.Context(W, H, bd)
        ctx.L.hop_cabac_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        for i in range(n):
            lg = 3 + (i % 4); cu = 1 << lg; n2 = cu * cu
            x, y = 64 * (i % 8), 64 * (i // 8)
            amp = float(rng.choice([4, 12, 40])) * (1 << (bd - 8))
            yy, xx = np.mgrid[0:cu, 0:cu]
            ry = amp * np.sin(xx * rng.uniform(0.1, 0.9) + rng.uniform(0, 3)) * np.cos(yy * rng.uniform(0.1, 0.9)) + rng.normal(0, amp / 4, (cu, cu))
            if rng.random() < 0.5:                                                     # spikes: transform skip territory
                m = rng.random((cu, cu)) < 0.06; ry[m] += rng.choice([-1, 1], int(m.sum())) * 6 * amp
            ry = np.clip(np.rint(ry), -(1 << bd) + 1, (1 << bd) - 1).astype(np.int16)
            rc = [np.clip(np.rint(ry[::2, ::2] * s + rng.normal(0, amp / 6, (cu // 2, cu // 2))), -(1 << bd) + 1, (1 << bd) - 1).astype(np.int16) for s in (0.5, -0.4)]
            resis.append(np.concatenate([ry.ravel(), rc[0].ravel(), rc[1].ravel()]))
            org[0][y:y + cu, x:x + cu] += ry; org[1][y // 2:(y + cu) // 2, x // 2:(x + cu) // 2] += rc[0]; org[2][y // 2:(y + cu) // 2, x // 2:(x + cu) // 2] += rc[1]
            qp = int(rng.integers(20, 43)) + 6 * (bd - 8); lam = 0.57 * 2.0 ** ((qp - 6 * (bd - 8) - 12) / 3.0); w = float(rng.choice([1.0, 1.26, 1.59]))
            c = cfgs[i]
            c["log2_cu"], c["qp"], c["bit_depth_y"], c["bit_depth_c"] = lg, (qp, qp - 1, qp - 2), bd, bd
            c["sign_hide"], c["use_ts"], c["log2_max_tu"] = int(rng.integers(0, 2)), int(rng.integers(0, 2)), 5
            c["inter_split_flag"] = int(rng.random() < 0.25)
            # getQuadtreeTULog2MinSizeInCU: three levels (QuadtreeTUMaxDepthInter 3), or one forced split (MaxDepthInter 1, partition != 2Nx2N)
            c["log2_min_tu_in_cu"] = min(lg - 1, 5) if c["inter_split_flag"] else max(2, lg - 2)
            c["lambda_rd"], c["lambda_rdoq"], c["dist_weight"] = lam, (lam, lam / w, lam / w), (1.0, w, w)
            j = jobs[i]
            j["x"], j["y"], j["log2_cu"], j["qp_scaled"], j["ctx_index"] = x, y, lg, c["qp"], i
            j["sign_hide"], j["use_ts"], j["log2_max_tu"], j["log2_min_tu_in_cu"], j["inter_split_flag"] = c["sign_hide"], c["use_ts"], 5, c["log2_min_tu_in_cu"], c["inter_split_flag"]
            j["lambda_rd"], j["lambda_rdoq"], j["dist_weight"] = lam, c["lambda_rdoq"], (w, w)
            assert ctx.L.hop_cabac_init(snaps[i].ctypes.data, int(rng.integers(0, 5)), int(rng.integers(20, 45))) == 0
            left = int(rng.integers(0, 32768)); snaps[i, 150], snaps[i, 151] = left & 255, left >> 8
        ctx.upload_orig(*org)
        for comp in range(3):
            ctx.plane_upload("pred", comp, np.full(org[comp].shape, 1 << (bd - 1), np.int16))
        res, co, cx = ctx.rqt(jobs, snaps)
        for comp in range(3):
            ctx.plane_upload("recon", comp, np.zeros(org[comp].shape, np.int16))
        res3, co3, fin3 = ctx.rqt_finish(jobs, res, co, cx)
        rec = [ctx.recon_download(k) for k in range(3)]
        off = 0; deep = ts = zeros = 0
        for i in range(n):
            cu = 1 << int(cfgs[i]["log2_cu"]); n2 = cu * cu; x, y = int(jobs[i]["x"]), int(jobs[i]["y"])
            o_i = np.concatenate([org[0][y:y + cu, x:x + cu].ravel(), org[1][y // 2:(y + cu) // 2, x // 2:(x + cu) // 2].ravel(), org[2][y // 2:(y + cu) // 2, x // 2:(x + cu) // 2].ravel()])
            want, arr, fin, (ocx, ofr), tail = oracle_rqt(cfgs[i], snaps[i, :150], int(snaps[i, 150]) | (int(snaps[i, 151]) << 8), resis[i], (o_i - resis[i]).astype(np.int16), o_i)
            # hop_rqt_finish: root-cbf-zero test, reconstruction, final distortions, cleared arrays / levels
            got_rec = np.concatenate([rec[0][y:y + cu, x:x + cu].ravel(), rec[1][y // 2:(y + cu) // 2, x // 2:(x + cu) // 2].ravel(), rec[2][y // 2:(y + cu) // 2, x // 2:(x + cu) // 2].ravel()])
            assert int(fin3[i, 0]) == tail["root"] and [int(v) for v in fin3[i, 1:]] == tail["d3"] and np.array_equal(got_rec, tail["rec"]), (bd, i, cfgs[i], fin3[i], tail["root"], tail["d3"])
            g3 = np.concatenate([res3[i]["tr_idx"][None, :], res3[i]["cbf"], res3[i]["tskip"]])
            assert np.array_equal(g3[:, :n2 // 16], tail["arr"][:, :n2 // 16]) and np.array_equal(co3[off:off + n2 * 3 // 2], tail["fin"]), (bd, i)
            zeros += int(tail["root"] == 0)
            cu = 1 << int(cfgs[i]["log2_cu"]); m = cu * cu * 3 // 2; parts = cu * cu // 16
            r = res[i]
            assert (float(r["cost"]), int(r["bits"]), int(r["dist"]), int(r["zero_dist"])) == want, (bd, i, cfgs[i], r["cost"], r["bits"], r["dist"], want)
            got = np.concatenate([r["tr_idx"][None, :], r["cbf"], r["tskip"]])
            assert np.array_equal(got[:, :parts], arr[:, :parts]) and np.array_equal(co[off:off + m], fin), (bd, i, cfgs[i])
            assert np.array_equal(cx[i, :150], ocx) and (int(cx[i, 150]) | (int(cx[i, 151]) << 8)) == ofr, (bd, i)
            deep += int(arr[0, :parts].max() >= 2); ts += int(arr[4:, :parts].any()); off += m
        assert deep >= 2 and ts >= 2 and zeros >= 1, (bd, deep, ts, zeros)
        ctx.close()


def test_cu_bits_encoder_calls(hp):
    """hop_inter_cu_bits (first piece of row a0: the CU-level syntax of an SS/GT CU through the lane-wise counting coder) on 57 xAddSymbolBitsInter calls
    recorded inside the encoder: every partition shape incl. AMP, merge / skip cases, MVDs, GT flags and vectors, transform trees with and without
    transform skip: bits, the skip decision, all residual and CU-level context states afterwards"""
    from goldutil import encoder_cubits_calls
    cases = list(encoder_cubits_calls())
    n = len(cases)
    jobs = np.zeros(n, hp.RQT_JOB_DTYPE); syn = np.zeros(n, hp.CU_SYNTAX_DTYPE); res = np.zeros(n, hp.RQT_RESULT_DTYPE)
    snaps = np.zeros((n, hp.CABAC_CTX_BYTES), np.uint8); cus = np.zeros((n, hp.CABAC_CU_CTX_BYTES), np.uint8)
    for i, c in enumerate(cases):
        cfg = c["cfg"]; j = jobs[i]
        j["log2_cu"], j["ctx_index"], j["sign_hide"], j["use_ts"], j["log2_max_tu"], j["log2_min_tu_in_cu"], j["inter_split_flag"] = (
            int(cfg["log2_cu"]), i, cfg["sign_hide"], cfg["use_ts"], cfg["log2_max_tu"], cfg["log2_min_tu_in_cu"], cfg["inter_split_flag"])
        j["lambda_rd"] = 1.0; j["lambda_rdoq"] = 1.0
        syn[i] = c["syn"]
        a = c["arr"].reshape(7, 256); res[i]["tr_idx"] = a[0]; res[i]["cbf"] = a[1:4]; res[i]["tskip"] = a[4:7]
        snaps[i, :150] = c["cin"]["ctx"]; left = int(c["cin"]["frac"]) & 32767; snaps[i, 150], snaps[i, 151] = left & 255, left >> 8
        cus[i, :16] = c["cuin"]
    ctx = hp.Context(64, 64)
    bits, sk, cx, cu = ctx.inter_cu_bits(jobs, syn, res, np.concatenate([c["coef"] for c in cases]), snaps, cus)
    for i, c in enumerate(cases):
        assert (int(bits[i]), int(sk[i])) == (c["bits"], c["skipped"]), (i, int(c["cfg"]["log2_cu"]), int(c["syn"]["part_size"]), bits[i], c["bits"])
        assert np.array_equal(cx[i, :150], c["cout"]["ctx"]) and (int(cx[i, 150]) | (int(cx[i, 151]) << 8)) == (int(c["cout"]["frac"]) & 32767), i
        assert np.array_equal(cu[i, :16], c["cuout"]) and not cu[i, 16:].any(), i
    bad = syn.copy(); bad[0]["part_size"] = 9
    with pytest.raises(hp.HopError):
        ctx.inter_cu_bits(jobs, bad, res, np.concatenate([c["coef"] for c in cases]), snaps, cus)
    ctx.close()


def test_intra_modes_vs_oracle(hp):
    """hop_intra_modes (mode bits + cost + candidate list + MPM append of the intra rough search) against the restatement, whose pieces stand in for
    xModeBitsIntra / xUpdateCandList inside the reference encoder: 4000 random blocks, both list sizes, 0..3 most probable modes, ties in the costs"""
    import ctypes
    from hoputil import oracle
    O = oracle()
    O.hop_o_intra_cand_list.restype = ctypes.c_int
    O.hop_o_intra_cand_list.argtypes = [ctypes.c_void_p, ctypes.c_uint8, ctypes.c_uint32, ctypes.c_double, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rng = np.random.default_rng(12)
    n = 4000
    jobs = np.zeros(n, hp.INTRA_MODES_JOB_DTYPE)
    satd = rng.integers(0, 5000, (n, 35)).astype(np.uint32)
    satd[::7] = rng.integers(0, 12, (len(satd[::7]), 35))                       # many equal costs: the insertion order decides
    for i in range(n):
        j = jobs[i]
        pn = int(rng.integers(0, 4)); j["pred_num"] = pn
        j["preds"] = [-1, -1, -1]; j["preds"][:pn] = rng.permutation(35)[:pn]
        j["mpm_cand"] = int(rng.integers(0, pn + 1)); j["num_full_rd"] = int(rng.choice([3, 8]))
        j["ctx_state"], j["frac_left"], j["sqrt_lambda"] = int(rng.integers(0, 126)), int(rng.integers(0, 32768)), float(rng.choice([7.6097, 4.8, 12.08]))
    ctx = hp.Context(64, 64)
    res = ctx.intra_modes(jobs, satd)
    for i in range(n):
        j = jobs[i]
        modes = (ctypes.c_uint32 * 11)(); costs = (ctypes.c_double * 8)(); preds = (ctypes.c_int * 3)(*[int(v) for v in j["preds"]])
        cnt = O.hop_o_intra_cand_list(satd[i].ctypes.data, int(j["ctx_state"]), int(j["frac_left"]), float(j["sqrt_lambda"]), preds, int(j["pred_num"]), int(j["mpm_cand"]),
                                      int(j["num_full_rd"]), modes, costs)
        assert int(res[i]["n"]) == cnt and list(res[i]["modes"][:cnt]) == list(modes)[:cnt] and list(res[i]["costs"][:int(j["num_full_rd"])]) == list(costs)[:int(j["num_full_rd"])], i
    bad = jobs[:1].copy(); bad["num_full_rd"] = 9
    with pytest.raises(hp.HopError):
        ctx.intra_modes(bad, satd[:1])
    ctx.close()


def test_intra_cu_bits_encoder_calls(hp):
    """hop_intra_cu_bits (xGetIntraBitsQT through the lane-wise counting coder: intra CU header with the luma directions against their most probable modes, chroma
    direction, split / cbf tree, levels with the direction-dependent scans) on 87 calls recorded inside the encoder: 2Nx2N and NxN, every node depth, luma-only /
    chroma-only / both: bits and all residual and CU-level context states afterwards"""
    from goldutil import encoder_intrabits_calls
    cases = list(encoder_intrabits_calls())
    n = len(cases)
    jobs = np.zeros(n, hp.RQT_JOB_DTYPE); syn = np.zeros(n, hp.INTRA_CU_SYNTAX_DTYPE); res = np.zeros(n, hp.RQT_RESULT_DTYPE)
    snaps = np.zeros((n, hp.CABAC_CTX_BYTES), np.uint8); cus = np.zeros((n, hp.CABAC_CU_CTX_BYTES), np.uint8)
    for i, c in enumerate(cases):
        cfg = c["cfg"]; j = jobs[i]
        j["log2_cu"], j["ctx_index"], j["sign_hide"], j["use_ts"], j["log2_max_tu"], j["log2_min_tu_in_cu"] = int(cfg["log2_cu"]), i, cfg["sign_hide"], cfg["use_ts"], cfg["log2_max_tu"], cfg["log2_min_tu_in_cu"]
        j["lambda_rd"] = 1.0; j["lambda_rdoq"] = 1.0
        for k in ("part_nxn", "skip_flag", "skip_ctx", "is_min_cu", "luma_dir", "preds", "pred_num", "chroma_is_dm", "chroma_dir"): syn[i][k] = c["syn"][k]
        syn[i]["tr_depth"], syn[i]["part"], syn[i]["b_luma"], syn[i]["b_chroma"] = c["nd"]
        a = c["arr"].reshape(7, 256); res[i]["tr_idx"] = a[0]; res[i]["cbf"] = a[1:4]; res[i]["tskip"] = a[4:7]
        snaps[i, :150] = c["cin"]["ctx"]; left = int(c["cin"]["frac"]) & 32767; snaps[i, 150], snaps[i, 151] = left & 255, left >> 8
        cus[i] = c["cuin"]
    ctx = hp.Context(64, 64)
    bits, cx, cu = ctx.intra_cu_bits(jobs, syn, res, np.concatenate([c["coef"] for c in cases]), snaps, cus)
    for i, c in enumerate(cases):
        assert int(bits[i]) == c["bits"], (i, int(c["cfg"]["log2_cu"]), c["nd"], int(bits[i]), c["bits"])
        assert np.array_equal(cx[i, :150], c["cout"]["ctx"]) and (int(cx[i, 150]) | (int(cx[i, 151]) << 8)) == (int(c["cout"]["frac"]) & 32767), i
        assert np.array_equal(cu[i], c["cuout"]), i
    bad = syn.copy(); bad[0]["part"] = 3
    with pytest.raises(hp.HopError):
        ctx.intra_cu_bits(jobs, bad, res, np.concatenate([c["coef"] for c in cases]), snaps, cus)
    ctx.close()


def test_intra_pred_chroma_vs_oracle(hp):
    """hop_intra_pred_chroma (initAdiPatternChroma + predIntraChromaAng for both planes) against the restatement that stands in for fillReferenceSamples /
    predIntraChromaAng inside the reference encoder: every chroma block size, all 35 directions, random availability per 2-sample unit, picture borders"""
    O = oracle()
    rng = np.random.default_rng(21)
    W, H = 256, 192
    Y, Cb, Cr = lenslet(W, H, 15, 3)
    rec = [Y, (Cb + rng.integers(-3, 4, Cb.shape)).clip(0, 255).astype(np.int16), (Cr + rng.integers(-3, 4, Cr.shape)).clip(0, 255).astype(np.int16)]
    ctx = hp.Context(W, H)
    ctx.upload_orig(Y, Cb, Cr)
    for k in range(3):
        ctx.plane_upload("recon", k, rec[k])
    n = 0
    for N in (4, 8, 16, 32):
        for trial in range(40):
            cx, cy = int(rng.integers(0, (W // 2 - N) // 4 + 1)) * 4, int(rng.integers(0, (H // 2 - N) // 4 + 1)) * 4
            U = N // 2
            fl = np.zeros(68, np.uint8)
            for u in range(4 * U + 1):
                if u < 2 * U: ok = cx > 0 and cy + 2 * (2 * U - 1 - u) + 2 <= H // 2
                elif u == 2 * U: ok = cx > 0 and cy > 0
                else: ok = cy > 0 and cx + 2 * (u - 2 * U - 1) + 2 <= W // 2
                fl[u] = int(ok and rng.random() < 0.8)
            mode = int(rng.integers(0, 35)) if trial >= 35 else trial
            j = hp.IntraJob(2 * cx, 2 * cy, N, 0)
            for k in range(68): j.flags[k] = int(fl[k])
            ctx.intra_pred_chroma([j], [mode])
            for comp in (1, 2):
                got = ctx.pred_download(comp)[cy:cy + N, cx:cx + N]
                L = (ctypes.c_int * (4 * 64 + 8))()
                O.hop_o_intra_fill_refs_u(p16(rec[comp]), W // 2, cx, cy, N, 2, fl.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), 8, L)
                want = np.zeros((N, N), np.int16)
                O.hop_o_intra_pred_chroma(L, N, mode, 8, p16(want))
                assert np.array_equal(got, want), (N, cx, cy, mode, comp)
            n += 1
    assert n == 160
    bad = hp.IntraJob(4, 0, 8, 0)
    with pytest.raises(hp.HopError):
        ctx.intra_pred_chroma([bad], [3])
    ctx.close()


def test_tu_intra_transform_skip_vs_oracle(hp):
    """the intra leaf with the 4x4 transform-skip variant (xIntraCodingLumaBlk / ChromaBlk with getTransformSkip set, as xRecurIntraCodingQT's retry calls it):
    hop_tu_rd with is_intra + HOP_TU_RD_TS against the restatement (whose members stand in for the reference's inside the encoder): levels, bits, distortion,
    cost, reconstruction"""
    O = oracle()
    O.hop_o_tu_intra_ts.restype = ctypes.c_int
    rng = np.random.default_rng(31)
    W = H = 128
    n = 200
    org = [rng.integers(0, 256, (H, W)).astype(np.int16), rng.integers(0, 256, (H // 2, W // 2)).astype(np.int16), rng.integers(0, 256, (H // 2, W // 2)).astype(np.int16)]
    prd = [np.clip(o + rng.integers(-4, 5, o.shape) + (rng.random(o.shape) < 0.1) * rng.integers(-90, 91, o.shape), 0, 255).astype(np.int16) for o in org]
    jobs = np.zeros(n, hp.TU_RD_JOB_DTYPE); snaps = np.zeros((n, hp.CABAC_CTX_BYTES), np.uint8)
    ctx = hp.Context(W, H)
    ctx.L.hop_cabac_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    used = set()
    for i in range(n):
        comp = int(rng.integers(0, 3))
        while True:
            bx, by = int(rng.integers(0, 16)), int(rng.integers(0, 16))
            if (comp, bx, by) not in used: used.add((comp, bx, by)); break
        j = jobs[i]
        j["x"], j["y"] = (4 * bx, 4 * by) if comp == 0 else (8 * bx, 8 * by)
        qp = int(rng.integers(22, 40)); lam = 0.57 * 2.0 ** ((qp - 12) / 3.0)
        j["comp"], j["log2_size"], j["qp_scaled"], j["tr_depth"], j["ctx_index"], j["sign_hide"], j["use_ts"], j["bit_depth"] = comp, 2, qp, int(rng.integers(0, 3)), i, int(rng.integers(0, 2)), 1, 8
        j["is_intra"], j["scan_idx"], j["use_dst"], j["flags"] = 1, int(rng.integers(0, 3)), 1, hp.HOP_TU_RD_TS if i % 4 else 0
        j["lambda_rdoq"], j["lambda_rd"], j["dist_weight"] = lam, lam, (1.0 if comp == 0 else 1.26)
        assert ctx.L.hop_cabac_init(snaps[i].ctypes.data, int(rng.integers(0, 5)), qp) == 0
        left = int(rng.integers(0, 32768)); snaps[i, 150], snaps[i, 151] = left & 255, left >> 8
    ctx.upload_orig(*org)
    for comp in range(3):
        ctx.plane_upload("pred", comp, prd[comp]); ctx.plane_upload("recon", comp, np.zeros(org[comp].shape, np.int16))
    res, lv = ctx.tu_rd(jobs, snaps)
    rec = [ctx.recon_download(k) for k in range(3)]
    nz = 0
    for i in range(n):
        j = jobs[i]; comp = int(j["comp"]); x, y = (int(j["x"]), int(j["y"])) if comp == 0 else (int(j["x"]) // 2, int(j["y"]) // 2)
        o = np.ascontiguousarray(org[comp][y:y + 4, x:x + 4]); p = np.ascontiguousarray(prd[comp][y:y + 4, x:x + 4])
        levels = np.zeros(16, np.int32); recon = np.zeros(16, np.int16); out = (ctypes.c_uint32 * 8)(); cost = ctypes.c_double()
        st = (ctypes.c_uint8 * 150)(*snaps[i, :150].tolist())
        O.hop_o_tu_intra_ts(p16(o), p16(p), 2, comp, int(j["scan_idx"]), 1, int(j["qp_scaled"]), 8, int(j["tr_depth"]), int(j["sign_hide"]), 1, int(bool(j["flags"])),
                            ctypes.c_double(float(j["lambda_rdoq"])), ctypes.c_double(float(j["lambda_rd"])), ctypes.c_double(float(j["dist_weight"])), st,
                            ctypes.c_uint32(int(snaps[i, 150]) | (int(snaps[i, 151]) << 8)), levels.ctypes.data_as(ctypes.c_void_p), p16(recon), out, ctypes.byref(cost))
        got = [int(res[i][k]) for k in ("abs_sum", "cbf", "dist", "bits")]
        assert got == [out[0], out[1], out[2], out[5]] and float(res[i]["cost"]) == cost.value, (i, comp, int(j["flags"]), got, list(out))
        assert np.array_equal(lv[16 * i:16 * i + 16], levels) and np.array_equal(rec[comp][y:y + 4, x:x + 4].ravel(), recon), (i, comp, int(j["flags"]))
        nz += int(out[0] != 0 and int(j["flags"]) != 0)
    assert nz > 60
    ctx.close()


def test_intra_rqt_encoder_calls(hp):
    """hop_intra_rqt (xRecurIntraCodingQT, luma: the transform tree of an intra PU - prediction from the reconstruction picture, leaf step, 4x4 transform-skip retry,
    bits via xGetIntraBitsQT, children on the state the previous one left, recount, split decision, the picture restored where the single block wins) on the 47 calls
    recorded inside the encoder, each in its own tile of one picture (its recorded neighbourhood around it): cost, distortion, depth / cbf / transform-skip arrays,
    chosen levels, coder and CU-level context states, and the reconstruction picture afterwards"""
    from goldutil import encoder_irqt_calls
    cases = list(encoder_irqt_calls())
    n = len(cases); T = 192; G = 7; W = H = T * G
    assert n <= G * G
    Y = np.zeros((H, W), np.int16); R = np.zeros((H, W), np.int16)
    jobs = np.zeros(n, hp.RQT_JOB_DTYPE); syn = np.zeros(n, hp.INTRA_CU_SYNTAX_DTYPE); opts = np.zeros(n, hp.INTRA_RQT_OPT_DTYPE)
    snaps = np.zeros((n, hp.CABAC_CTX_BYTES), np.uint8); cus = np.zeros((n, hp.CABAC_CU_CTX_BYTES), np.uint8)
    for i, c in enumerate(cases):
        cfg = c["cfg"]; j = jobs[i]; cu = 1 << int(cfg["log2_cu"]); Wn = 2 * cu + 1
        x0, y0 = (i % G) * T + 64, (i // G) * T + 64
        Y[y0:y0 + cu, x0:x0 + cu] = c["org"].reshape(cu, cu)
        R[y0 - 1:y0 - 1 + Wn, x0 - 1:x0 - 1 + Wn] = c["win"].reshape(Wn, Wn)
        j["x"], j["y"], j["log2_cu"], j["qp_scaled"], j["ctx_index"] = x0, y0, int(cfg["log2_cu"]), cfg["qp"], i
        j["sign_hide"], j["use_ts"], j["log2_max_tu"], j["log2_min_tu_in_cu"] = cfg["sign_hide"], cfg["use_ts"], cfg["log2_max_tu"], cfg["log2_min_tu_in_cu"]
        j["lambda_rd"], j["lambda_rdoq"], j["dist_weight"] = cfg["lambda_rd"], cfg["lambda_rdoq"], cfg["dist_weight"][1:]
        for k in ("part_nxn", "skip_flag", "skip_ctx", "is_min_cu", "luma_dir", "preds", "pred_num", "chroma_is_dm", "chroma_dir"): syn[i][k] = c["syn"][k]
        syn[i]["tr_depth"], syn[i]["part"], syn[i]["b_luma"] = c["nd"][0], c["nd"][1], 1
        opts[i]["check_first"], opts[i]["ts_fast"], opts[i]["strong"] = c["nd"][2], c["nd"][3], c["nd"][4]
        av = c["avail"].reshape(341, 36).astype(np.uint64)
        opts[i]["avail"] = (av << np.arange(36, dtype=np.uint64)[None, :]).sum(axis=1)
        snaps[i, :150] = c["cin"]["ctx"]; left = int(c["cin"]["frac"]) & 32767; snaps[i, 150], snaps[i, 151] = left & 255, left >> 8
        cus[i] = c["cuin"]
    ctx = hp.Context(W, H)
    ctx.upload_orig(Y, np.zeros((H // 2, W // 2), np.int16), np.zeros((H // 2, W // 2), np.int16))
    ctx.plane_upload("recon", 0, R)
    res, coef, cx, cu_out = ctx.intra_rqt(jobs, syn, opts, snaps, cus)
    R2 = ctx.recon_download(0)
    o = 0; kinds = set()
    for i, c in enumerate(cases):
        cu = 1 << int(c["cfg"]["log2_cu"]); parts = (cu // 4) ** 2; p0 = c["nd"][1]; np_ = parts >> (2 * c["nd"][0])
        tag = (i, cu, c["nd"][:3])
        assert float(res[i]["cost"]) == c["cost"] and int(res[i]["dist"]) == c["dist"], (tag, float(res[i]["cost"]), c["cost"], int(res[i]["dist"]), c["dist"])
        a = c["aout"].reshape(7, 256)
        assert np.array_equal(res[i]["tr_idx"][p0:p0 + np_], a[0, p0:p0 + np_]) and np.array_equal(res[i]["cbf"][0][p0:p0 + np_], a[1, p0:p0 + np_]), tag
        assert np.array_equal(res[i]["tskip"][0][p0:p0 + np_], a[4, p0:p0 + np_]), tag
        assert np.array_equal(coef[o + 16 * p0:o + 16 * (p0 + np_)], c["fin"][16 * p0:16 * (p0 + np_)]), tag
        assert np.array_equal(cx[i, :150], c["cout"]["ctx"]) and (int(cx[i, 150]) | (int(cx[i, 151]) << 8)) == (int(c["cout"]["frac"]) & 32767), tag
        assert np.array_equal(cu_out[i], c["cuout"]), tag
        x0, y0 = (i % G) * T + 64, (i // G) * T + 64
        assert np.array_equal(R2[y0:y0 + cu, x0:x0 + cu].reshape(-1), c["rec"]), tag
        R[y0:y0 + cu, x0:x0 + cu] = c["rec"].reshape(cu, cu)
        kinds.add((cu, c["nd"][0], c["nd"][2], int(a[0, p0:p0 + np_].max()), int(a[4, p0:p0 + np_].any())))
        o += cu * cu * 3 // 2
    assert np.array_equal(R, R2)                                       # nothing outside the PUs' CUs was touched
    assert len(kinds) >= 12 and any(k[4] for k in kinds)
    bad = syn.copy(); bad[0]["part"] = 3
    with pytest.raises(hp.HopError):
        ctx.intra_rqt(jobs, bad, opts, snaps, cus)
    ctx.close()


def _irqt_random_cases(hp, L, bd):
    """72 random PUs, each in its own 192 x 192 tile (CU at +64): pictures, jobs, syntax, options, restatement configs, snapshots, availability tables"""
    from goldutil import RQT_CFG
    rng = np.random.default_rng(70 + bd)
    T = 192; G = 9; W = H = T * G; n = 72
    mid = 1 << (bd - 1); top = (1 << bd) - 1
    Y = np.full((H, W), mid, np.int16)
    R = rng.integers(0, top + 1, (H, W)).astype(np.int16)          # whatever lies around (and, before it is coded, inside) the CUs
    jobs = np.zeros(n, hp.RQT_JOB_DTYPE); syn = np.zeros(n, hp.INTRA_CU_SYNTAX_DTYPE); opts = np.zeros(n, hp.INTRA_RQT_OPT_DTYPE); cfgs = np.zeros(n, RQT_CFG)
    snaps = np.zeros((n, hp.CABAC_CTX_BYTES), np.uint8); cus = np.zeros((n, hp.CABAC_CU_CTX_BYTES), np.uint8); avs = []
    L.hop_cabac_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]; L.hop_cabac_cu_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    for i in range(n):
        lg = 3 + (i % 4); cu = 1 << lg
        nxn = int(lg == 3 and rng.random() < 0.6)
        x0, y0 = (i % G) * T + 64, (i // G) * T + 64
        # a smooth random field over the CU and its neighbourhood (references further away predict worse: splitting pays), the picture around the CU = the field plus
        # coding noise; on half of the PUs isolated spikes on top (transform-skip territory)
        amp = float(rng.choice([20, 50, 90])) * (1 << (bd - 8)); Wn = 2 * cu + 1
        f = gaussian_filter(rng.normal(0, 1, (Wn + 16, Wn + 16)), float(rng.choice([1.5, 3, 6])))[8:-8, 8:-8]
        f = mid + amp * f / max(1e-9, float(np.abs(f).max()))
        if rng.random() < 0.5:
            m = rng.random(f.shape) < 0.05; f[m] += rng.choice([-1, 1], int(m.sum())) * 2.5 * amp
        f = np.clip(np.rint(f), 0, top).astype(np.int16)
        Y[y0:y0 + cu, x0:x0 + cu] = f[1:1 + cu, 1:1 + cu]
        R[y0 - 1:y0 + 2 * cu, x0 - 1:x0 + 2 * cu] = np.clip(f + rng.integers(-2, 3, f.shape), 0, top)
        qp = int(rng.integers(20, 43)) + 6 * (bd - 8); lam = 0.57 * 2.0 ** ((qp - 6 * (bd - 8) - 12) / 3.0)
        c = cfgs[i]
        c["log2_cu"], c["qp"], c["bit_depth_y"], c["bit_depth_c"] = lg, (qp, qp, qp), bd, bd
        c["sign_hide"], c["use_ts"], c["log2_max_tu"] = int(rng.integers(0, 2)), int(rng.random() < 0.7), 5
        c["log2_min_tu_in_cu"] = max(2, min(lg - nxn, 5) - int(rng.integers(0, 3)))
        c["lambda_rd"], c["lambda_rdoq"], c["dist_weight"] = lam, (lam, lam, lam), (1.0, 1.0, 1.0)
        j = jobs[i]
        j["x"], j["y"], j["log2_cu"], j["qp_scaled"], j["ctx_index"] = x0, y0, lg, c["qp"], i
        j["sign_hide"], j["use_ts"], j["log2_max_tu"], j["log2_min_tu_in_cu"] = c["sign_hide"], c["use_ts"], 5, c["log2_min_tu_in_cu"]
        j["lambda_rd"], j["lambda_rdoq"], j["dist_weight"] = lam, c["lambda_rdoq"], (1.0, 1.0)
        s = syn[i]
        s["part_nxn"], s["skip_flag"], s["skip_ctx"], s["is_min_cu"] = nxn, 0, int(rng.integers(0, 3)), int(lg == 3)
        s["luma_dir"] = rng.integers(0, 35, 4)
        for p in range(4):
            s["preds"][p] = rng.choice(35, 3, replace=False); s["pred_num"][p] = 3
            if rng.random() < 0.4: s["luma_dir"][p] = s["preds"][p][int(rng.integers(0, 3))]
        s["chroma_is_dm"], s["chroma_dir"] = 1, 36
        s["tr_depth"], s["part"], s["b_luma"] = nxn, (int(rng.integers(0, 4)) * ((cu // 4) ** 2 // 4) if nxn else 0), 1
        opts[i]["check_first"], opts[i]["ts_fast"], opts[i]["strong"] = int(rng.random() < 0.4), int(rng.random() < 0.3), int(rng.integers(0, 2))
        av = (rng.random((341, 36)) < (0.85 if rng.random() < 0.8 else 0.3)).astype(np.uint8); avs.append(av)
        opts[i]["avail"] = (av.astype(np.uint64) << np.arange(36, dtype=np.uint64)[None, :]).sum(axis=1)
        assert L.hop_cabac_init(snaps[i].ctypes.data, int(rng.integers(0, 5)), int(rng.integers(20, 45))) == 0
        assert L.hop_cabac_cu_init(cus[i].ctypes.data, int(rng.integers(0, 5)), int(rng.integers(20, 45))) == 0
        left = int(rng.integers(0, 32768)); snaps[i, 150], snaps[i, 151] = left & 255, left >> 8
    return W, H, Y, R, jobs, syn, opts, cfgs, snaps, cus, avs


def test_intra_rqt_random_vs_oracle(hp):
    """hop_intra_rqt against the restatement (pinned inside the reference encoder) where the recorded calls do not go: 8 and 10 bit, QP 20..42, sign hiding / transform
    skip / TransformSkipFast / strong smoothing on and off, every CU size, 2Nx2N and the four PUs of NxN, bCheckFirst on and off, deeper trees than the encoder's
    configuration allows, random neighbour availability per node, smooth and sharp content, context states of all slice types with a carried fraction.  72 PUs per bit
    depth in one call (several classes), each in its own tile."""
    import ctypes
    from goldutil import oracle_intra_rqt, RQT_CFG, INTRA_SYN
    for bd in (8, 10):
        W, H, Y, R, jobs, syn, opts, cfgs, snaps, cus, avs = _irqt_random_cases(hp, hp.load(), bd)
        n = len(jobs); mid = 1 << (bd - 1)
        ctx = hp.Context(W, H, bd)
        ctx.upload_orig(Y, np.full((H // 2, W // 2), mid, np.int16), np.full((H // 2, W // 2), mid, np.int16))
        ctx.plane_upload("recon", 0, R)
        res, coef, cx, cu_out = ctx.intra_rqt(jobs, syn, opts, snaps, cus)
        R2 = ctx.recon_download(0)
        o = 0; deep = ts = single_over_split = 0
        for i in range(n):
            c = cfgs[i]; cu = 1 << int(c["log2_cu"]); Wn = 2 * cu + 1; parts = (cu // 4) ** 2
            x0, y0 = int(jobs[i]["x"]), int(jobs[i]["y"]); d0, p0 = int(syn[i]["tr_depth"]), int(syn[i]["part"]); np_ = parts >> (2 * d0)
            osyn = np.zeros(1, INTRA_SYN)
            for k in INTRA_SYN.names: osyn[0][k] = syn[i][k]
            coder = snaps[i, :150].tobytes() + b"\0\0" + (int(snaps[i, 150]) | (int(snaps[i, 151]) << 8)).to_bytes(8, "little")
            nd = [d0, p0, int(opts[i]["check_first"]), int(opts[i]["ts_fast"]), int(opts[i]["strong"])]
            cost, dist, arr, ocoder, ocu, win, fin = oracle_intra_rqt(c, osyn[0], nd, avs[i], Y[y0:y0 + cu, x0:x0 + cu], R[y0 - 1:y0 - 1 + Wn, x0 - 1:x0 - 1 + Wn], np.zeros(1792, np.uint8),
                                                                      np.frombuffer(coder, np.uint8), cus[i])
            tag = (bd, i, cu, nd, int(c["log2_min_tu_in_cu"]))
            assert float(res[i]["cost"]) == cost and int(res[i]["dist"]) == dist, (tag, float(res[i]["cost"]), cost, int(res[i]["dist"]), dist)
            assert np.array_equal(res[i]["tr_idx"][p0:p0 + np_], arr[0, p0:p0 + np_]) and np.array_equal(res[i]["cbf"][0][p0:p0 + np_], arr[1, p0:p0 + np_]), tag
            assert np.array_equal(res[i]["tskip"][0][p0:p0 + np_], arr[4, p0:p0 + np_]), tag
            assert np.array_equal(coef[o + 16 * p0:o + 16 * (p0 + np_)], fin[16 * p0:16 * (p0 + np_)]), tag
            assert np.array_equal(cx[i, :150], ocoder[:150]) and (int(cx[i, 150]) | (int(cx[i, 151]) << 8)) == (int.from_bytes(ocoder[152:160].tobytes(), "little") & 32767), tag
            assert np.array_equal(cu_out[i], ocu), tag
            R[y0 - 1:y0 - 1 + Wn, x0 - 1:x0 - 1 + Wn] = win.reshape(Wn, Wn)
            deep += int(arr[0, p0:p0 + np_].max() >= d0 + 2); ts += int(arr[4, p0:p0 + np_].any())
            lg = int(c["log2_cu"]); single_over_split += int(arr[0, p0] == d0 and lg - d0 > int(c["log2_min_tu_in_cu"]) and not nd[2] and lg - d0 <= 5)
            o += cu * cu * 3 // 2
        assert np.array_equal(R, R2), bd
        assert deep >= 4 and ts >= 3 and single_over_split >= 4, (bd, deep, ts, single_over_split)
        ctx.close()


def test_intra_luma_search_encoder_calls(hp):
    """hop_intra_luma_search (estIntraPredQT, luma: most probable modes, rough search, candidate list, candidate and final transform trees, the better result kept, the
    decided PU's block into the picture, NxN cbf combined) on the 43 calls recorded inside the encoder, each in its own tile of one picture: directions, candidates
    tested, distortion, arrays, the CU's luma levels, its reconstruction plane and the picture afterwards"""
    from goldutil import encoder_isearch_calls
    cases = list(encoder_isearch_calls())
    n = len(cases); T = 192; G = 7; W = H = T * G
    assert n <= G * G
    Y = np.zeros((H, W), np.int16); R = np.zeros((H, W), np.int16)
    jobs = np.zeros(n, hp.RQT_JOB_DTYPE); syn = np.zeros(n, hp.INTRA_CU_SYNTAX_DTYPE); opts = np.zeros(n, hp.INTRA_RQT_OPT_DTYPE); sj = np.zeros(n, hp.INTRA_SEARCH_JOB_DTYPE)
    snaps = np.zeros((n, hp.CABAC_CTX_BYTES), np.uint8); cus = np.zeros((n, hp.CABAC_CU_CTX_BYTES), np.uint8)
    for i, c in enumerate(cases):
        cfg = c["cfg"]; j = jobs[i]; cu = 1 << int(cfg["log2_cu"]); Wn = 2 * cu + 1
        x0, y0 = (i % G) * T + 64, (i // G) * T + 64
        Y[y0:y0 + cu, x0:x0 + cu] = c["org"].reshape(cu, cu)
        R[y0 - 1:y0 - 1 + Wn, x0 - 1:x0 - 1 + Wn] = c["win"].reshape(Wn, Wn)
        j["x"], j["y"], j["log2_cu"], j["qp_scaled"], j["ctx_index"] = x0, y0, int(cfg["log2_cu"]), cfg["qp"], i
        j["sign_hide"], j["use_ts"], j["log2_max_tu"], j["log2_min_tu_in_cu"] = cfg["sign_hide"], cfg["use_ts"], cfg["log2_max_tu"], cfg["log2_min_tu_in_cu"]
        j["lambda_rd"], j["lambda_rdoq"], j["dist_weight"] = cfg["lambda_rd"], cfg["lambda_rdoq"], cfg["dist_weight"][1:]
        for k in ("part_nxn", "skip_flag", "skip_ctx", "is_min_cu", "chroma_is_dm", "chroma_dir"): syn[i][k] = c["syn"][k]
        opts[i]["ts_fast"], opts[i]["strong"] = c["nd"][0], c["nd"][1]
        av = c["avail"].reshape(341, 36).astype(np.uint64)
        opts[i]["avail"] = (av << np.arange(36, dtype=np.uint64)[None, :]).sum(axis=1)
        sj[i]["left_dir"], sj[i]["above_dir"], sj[i]["rough_flags"], sj[i]["sqrt_lambda"], sj[i]["num_full_rd"] = c["dirs"][:4], c["dirs"][4:], c["rough"].reshape(4, 68), c["sql"], c["nd"][2]
        snaps[i, :150] = c["cin"]["ctx"]; left = int(c["cin"]["frac"]) & 32767; snaps[i, 150], snaps[i, 151] = left & 255, left >> 8
        cus[i] = c["cuin"]
    ctx = hp.Context(W, H)
    ctx.upload_orig(Y, np.zeros((H // 2, W // 2), np.int16), np.zeros((H // 2, W // 2), np.int16))
    ctx.plane_upload("recon", 0, R)
    sres, res, coef, reco = ctx.intra_luma_search(jobs, syn, opts, sj, snaps, cus)
    R2 = ctx.recon_download(0)
    o = ro = 0; kinds = set()
    for i, c in enumerate(cases):
        cu = 1 << int(c["cfg"]["log2_cu"]); parts = (cu // 4) ** 2; npu = 4 if c["syn"]["part_nxn"] else 1
        tag = (i, cu, npu)
        assert [int(v) for v in sres[i]["best_dir"][:npu]] == [int(v) for v in c["best"][:npu]], (tag, sres[i]["best_dir"], c["best"])
        assert [int(v) for v in sres[i]["n_cand"][:npu]] == [int(v) for v in c["ncand"][:npu]] and int(sres[i]["dist"]) == c["dist"], (tag, sres[i], c["ncand"], c["dist"])
        a = c["aout"].reshape(7, 256)
        assert np.array_equal(res[i]["tr_idx"][:parts], a[0, :parts]) and np.array_equal(res[i]["cbf"][0][:parts], a[1, :parts]) and np.array_equal(res[i]["tskip"][0][:parts], a[4, :parts]), tag
        assert np.array_equal(coef[o:o + cu * cu], c["coef"]) and np.array_equal(reco[ro:ro + cu * cu], c["reco"]), tag
        x0, y0 = (i % G) * T + 64, (i // G) * T + 64
        assert np.array_equal(R2[y0:y0 + cu, x0:x0 + cu].reshape(-1), c["rec"]), tag
        R[y0:y0 + cu, x0:x0 + cu] = c["rec"].reshape(cu, cu)
        kinds.add((cu, npu, int(c["ncand"][0]), int(a[0, :parts].max()), int(a[4, :parts].any())))
        o += cu * cu * 3 // 2; ro += cu * cu
    assert np.array_equal(R, R2)
    assert len(kinds) >= 12 and any(k[4] for k in kinds) and any(k[1] == 4 for k in kinds)
    bad = sj.copy(); bad[0]["left_dir"][0] = 35
    with pytest.raises(hp.HopError):
        ctx.intra_luma_search(jobs, syn, opts, bad, snaps, cus)
    ctx.close()


def test_intra_luma_search_random_vs_oracle(hp):
    """hop_intra_luma_search against the restatement (pinned inside the reference encoder) on the random CUs of test_intra_rqt_random_vs_oracle (8 and 10 bit, every CU size,
    NxN, deeper trees, random availability) with random neighbour directions and rough-search flags: 72 CUs per bit depth in one call (several classes)"""
    from goldutil import oracle_intra_luma_search, INTRA_SYN
    for bd in (8, 10):
        W, H, Y, R, jobs, syn, opts, cfgs, snaps, cus, avs = _irqt_random_cases(hp, hp.load(), bd)
        n = len(jobs); mid = 1 << (bd - 1)
        rng = np.random.default_rng(90 + bd)
        sj = np.zeros(n, hp.INTRA_SEARCH_JOB_DTYPE)
        sj["left_dir"] = rng.integers(0, 35, (n, 4)); sj["above_dir"] = rng.integers(0, 35, (n, 4))
        same = rng.random(n) < 0.3; sj["above_dir"][same] = sj["left_dir"][same]
        sj["rough_flags"] = rng.random((n, 4, 68)) < 0.8
        for i in range(n):
            sj[i]["sqrt_lambda"] = float(np.sqrt(jobs[i]["lambda_rd"])); sj[i]["num_full_rd"] = 8 if ((1 << int(jobs[i]["log2_cu"])) >> int(syn[i]["part_nxn"])) <= 8 else 3
        ctx = hp.Context(W, H, bd)
        ctx.upload_orig(Y, np.full((H // 2, W // 2), mid, np.int16), np.full((H // 2, W // 2), mid, np.int16))
        ctx.plane_upload("recon", 0, R)
        sres, res, coef, reco = ctx.intra_luma_search(jobs, syn, opts, sj, snaps, cus)
        R2 = ctx.recon_download(0)
        o = ro = 0; final_better = nxn_n = 0
        for i in range(n):
            c = cfgs[i]; cu = 1 << int(c["log2_cu"]); Wn = 2 * cu + 1; parts = (cu // 4) ** 2; npu = 4 if syn[i]["part_nxn"] else 1
            x0, y0 = int(jobs[i]["x"]), int(jobs[i]["y"])
            osyn = np.zeros(1, INTRA_SYN)
            for k in INTRA_SYN.names: osyn[0][k] = syn[i][k]
            coder = snaps[i, :150].tobytes() + b"\0\0" + (int(snaps[i, 150]) | (int(snaps[i, 151]) << 8)).to_bytes(8, "little")
            nd = [int(opts[i]["ts_fast"]), int(opts[i]["strong"]), int(sj[i]["num_full_rd"])]
            best, ncand, dist, arr, ocoef, oreco, win = oracle_intra_luma_search(c, osyn[0], nd, np.concatenate([sj[i]["left_dir"], sj[i]["above_dir"]]), float(sj[i]["sqrt_lambda"]),
                                                                                 sj[i]["rough_flags"], avs[i], Y[y0:y0 + cu, x0:x0 + cu], R[y0 - 1:y0 - 1 + Wn, x0 - 1:x0 - 1 + Wn],
                                                                                 np.frombuffer(coder, np.uint8), cus[i])
            tag = (bd, i, cu, npu)
            assert [int(v) for v in sres[i]["best_dir"][:npu]] == best[:npu] and [int(v) for v in sres[i]["n_cand"][:npu]] == ncand[:npu] and int(sres[i]["dist"]) == dist, (tag, sres[i], best, ncand, dist)
            assert np.array_equal(res[i]["tr_idx"][:parts], arr[0, :parts]) and np.array_equal(res[i]["cbf"][0][:parts], arr[1, :parts]) and np.array_equal(res[i]["tskip"][0][:parts], arr[4, :parts]), tag
            assert np.array_equal(coef[o:o + cu * cu], ocoef) and np.array_equal(reco[ro:ro + cu * cu], oreco), tag
            R[y0 - 1:y0 - 1 + Wn, x0 - 1:x0 - 1 + Wn] = win.reshape(Wn, Wn)
            final_better += int(not np.array_equal(oreco.reshape(cu, cu), win.reshape(Wn, Wn)[1:1 + cu, 1:1 + cu])); nxn_n += int(npu == 4)
            o += cu * cu * 3 // 2; ro += cu * cu
        assert np.array_equal(R, R2), bd
        assert nxn_n >= 4, (bd, nxn_n, final_better)
        ctx.close()


def test_intra_chroma_search_encoder_calls(hp):
    """hop_intra_chroma_search (estIntraPredChromaQT: the five allowed directions along the luma tree, chroma leaf per plane, transform-skip retry per component, the CU's
    chroma bits, the best kept) on the 44 calls recorded inside the encoder, each in its own tile: direction, distortion, cbf / transform-skip arrays, chroma levels,
    reconstruction planes and the chroma pictures afterwards"""
    from goldutil import encoder_csearch_calls
    cases = list(encoder_csearch_calls())
    n = len(cases); T = 192; G = 7; W = H = T * G
    assert n <= G * G
    C = [np.zeros((H // 2, W // 2), np.int16) for _ in range(2)]; R = [np.zeros((H // 2, W // 2), np.int16) for _ in range(2)]
    jobs = np.zeros(n, hp.RQT_JOB_DTYPE); syn = np.zeros(n, hp.INTRA_CU_SYNTAX_DTYPE); opts = np.zeros(n, hp.INTRA_RQT_OPT_DTYPE); res = np.zeros(n, hp.RQT_RESULT_DTYPE)
    snaps = np.zeros((n, hp.CABAC_CTX_BYTES), np.uint8); cus = np.zeros((n, hp.CABAC_CU_CTX_BYTES), np.uint8)
    for i, c in enumerate(cases):
        cfg = c["cfg"]; j = jobs[i]; cu = 1 << int(cfg["log2_cu"]); half = cu // 2; Wn = cu + 1
        x0, y0 = (i % G) * T + 64, (i // G) * T + 64
        for k in range(2):
            C[k][y0 // 2:y0 // 2 + half, x0 // 2:x0 // 2 + half] = c["org"].reshape(2, half, half)[k]
            R[k][y0 // 2 - 1:y0 // 2 - 1 + Wn, x0 // 2 - 1:x0 // 2 - 1 + Wn] = c["win"].reshape(2, Wn, Wn)[k]
        j["x"], j["y"], j["log2_cu"], j["qp_scaled"], j["ctx_index"] = x0, y0, int(cfg["log2_cu"]), cfg["qp"], i
        j["sign_hide"], j["use_ts"], j["log2_max_tu"], j["log2_min_tu_in_cu"] = cfg["sign_hide"], cfg["use_ts"], cfg["log2_max_tu"], cfg["log2_min_tu_in_cu"]
        j["lambda_rd"], j["lambda_rdoq"], j["dist_weight"] = cfg["lambda_rd"], cfg["lambda_rdoq"], cfg["dist_weight"][1:]
        for k in ("part_nxn", "skip_flag", "skip_ctx", "is_min_cu", "luma_dir", "preds", "pred_num", "chroma_is_dm", "chroma_dir"): syn[i][k] = c["syn"][k]
        opts[i]["ts_fast"] = c["nd"][0]
        av = c["avail"].reshape(341, 36).astype(np.uint64)
        opts[i]["avail"] = (av << np.arange(36, dtype=np.uint64)[None, :]).sum(axis=1)
        a = c["ain"].reshape(7, 256); res[i]["tr_idx"] = a[0]; res[i]["cbf"] = a[1:4]; res[i]["tskip"] = a[4:7]
        snaps[i, :150] = c["cin"]["ctx"]; left = int(c["cin"]["frac"]) & 32767; snaps[i, 150], snaps[i, 151] = left & 255, left >> 8
        cus[i] = c["cuin"]
    ctx = hp.Context(W, H)
    ctx.upload_orig(np.zeros((H, W), np.int16), C[0], C[1])
    for k in range(2): ctx.plane_upload("recon", 1 + k, R[k])
    cres, res2, coef, reco = ctx.intra_chroma_search(jobs, syn, opts, res, snaps, cus)
    R2 = [ctx.recon_download(1), ctx.recon_download(2)]
    o = ro = 0; kinds = set()
    for i, c in enumerate(cases):
        cu = 1 << int(c["cfg"]["log2_cu"]); half = cu // 2; parts = (cu // 4) ** 2
        tag = (i, cu, int(c["ain"][:parts].max()))
        assert int(cres[i]["best_mode"]) == c["mode"] and int(cres[i]["dist"]) == c["dist"], (tag, cres[i], c["mode"], c["dist"])
        a = c["aout"].reshape(7, 256)
        assert np.array_equal(res2[i]["cbf"][1:, :parts], a[2:4, :parts]) and np.array_equal(res2[i]["tskip"][1:, :parts], a[5:7, :parts]), tag
        assert np.array_equal(coef[o + cu * cu:o + cu * cu * 3 // 2], c["coef"]) and np.array_equal(reco[ro:ro + cu * cu // 2], c["reco"]), tag
        x0, y0 = (i % G) * T + 64, (i // G) * T + 64
        for k in range(2):
            blk = c["rec"].reshape(2, half, half)[k]
            assert np.array_equal(R2[k][y0 // 2:y0 // 2 + half, x0 // 2:x0 // 2 + half], blk), (tag, k)
            R[k][y0 // 2:y0 // 2 + half, x0 // 2:x0 // 2 + half] = blk
        kinds.add((cu, c["mode"], int(a[5:7, :parts].any()), int(a[0, :parts].max())))
        o += cu * cu * 3 // 2; ro += cu * cu // 2
    assert np.array_equal(R[0], R2[0]) and np.array_equal(R[1], R2[1])
    assert len(kinds) >= 15 and any(k[2] for k in kinds)
    bad = res.copy(); bad[0]["tr_idx"][0] = 7
    with pytest.raises(hp.HopError):
        ctx.intra_chroma_search(jobs, syn, opts, bad, snaps, cus)
    ctx.close()


def test_intra_chroma_search_random_vs_oracle(hp):
    """hop_intra_luma_search followed by hop_intra_chroma_search on the random CUs of the luma tests (8 and 10 bit, every CU size, NxN, deeper trees, random availability),
    with chroma planes of their own, chroma quantisers / lambdas / distortion weights that differ per plane, TransformSkipFast on and off: the chroma search against the
    restatement fed with the same luma arrays"""
    from goldutil import oracle_intra_chroma_search, INTRA_SYN
    for bd in (8, 10):
        W, H, Y, R, jobs, syn, opts, cfgs, snaps, cus, avs = _irqt_random_cases(hp, hp.load(), bd)
        n = len(jobs); mid = 1 << (bd - 1); top = (1 << bd) - 1
        rng = np.random.default_rng(110 + bd)
        sj = np.zeros(n, hp.INTRA_SEARCH_JOB_DTYPE)
        sj["left_dir"] = rng.integers(0, 35, (n, 4)); sj["above_dir"] = rng.integers(0, 35, (n, 4)); sj["rough_flags"] = 1
        C = [np.full((H // 2, W // 2), mid, np.int16) for _ in range(2)]; RC = [rng.integers(0, top + 1, (H // 2, W // 2)).astype(np.int16) for _ in range(2)]
        for i in range(n):
            cu = 1 << int(jobs[i]["log2_cu"]); half = cu // 2; Wn = cu + 1; x0, y0 = int(jobs[i]["x"]) // 2, int(jobs[i]["y"]) // 2
            sj[i]["sqrt_lambda"] = float(np.sqrt(jobs[i]["lambda_rd"])); sj[i]["num_full_rd"] = 8 if (cu >> int(syn[i]["part_nxn"])) <= 8 else 3
            for k in range(2):
                amp = float(rng.choice([15, 40, 80])) * (1 << (bd - 8))
                f = gaussian_filter(rng.normal(0, 1, (Wn + 16, Wn + 16)), float(rng.choice([1.0, 2.5, 5])))[8:-8, 8:-8]
                f = mid + amp * f / max(1e-9, float(np.abs(f).max()))
                if rng.random() < 0.5:
                    m = rng.random(f.shape) < 0.06; f[m] += rng.choice([-1, 1], int(m.sum())) * 2.5 * amp
                f = np.clip(np.rint(f), 0, top).astype(np.int16)
                C[k][y0:y0 + half, x0:x0 + half] = f[1:1 + half, 1:1 + half]
                RC[k][y0 - 1:y0 - 1 + Wn, x0 - 1:x0 - 1 + Wn] = np.clip(f + rng.integers(-2, 3, f.shape), 0, top)
            qp = int(jobs[i]["qp_scaled"][0]); w = (float(rng.choice([1.0, 1.26, 1.59])), float(rng.choice([1.0, 1.26])))
            jobs[i]["qp_scaled"] = (qp, max(6 * (bd - 8), qp - int(rng.integers(0, 4))), max(6 * (bd - 8), qp - int(rng.integers(0, 6))))
            lam = float(jobs[i]["lambda_rd"]); jobs[i]["lambda_rdoq"] = (lam, lam / w[0], lam / w[1]); jobs[i]["dist_weight"] = w
            cfgs[i]["qp"] = jobs[i]["qp_scaled"]; cfgs[i]["lambda_rdoq"] = jobs[i]["lambda_rdoq"]; cfgs[i]["dist_weight"] = (1.0, w[0], w[1])
        ctx = hp.Context(W, H, bd)
        ctx.upload_orig(Y, C[0], C[1])
        ctx.plane_upload("recon", 0, R)
        for k in range(2): ctx.plane_upload("recon", 1 + k, RC[k])
        sres, res, coef_y, reco_y = ctx.intra_luma_search(jobs, syn, opts, sj, snaps, cus)
        syn2 = syn.copy(); syn2["luma_dir"] = sres["best_dir"]
        cres, res2, coef, reco = ctx.intra_chroma_search(jobs, syn2, opts, res, snaps, cus)
        R2 = [ctx.recon_download(1), ctx.recon_download(2)]
        o = ro = 0; ts = 0; modes = set()
        for i in range(n):
            c = cfgs[i]; cu = 1 << int(c["log2_cu"]); half = cu // 2; Wn = cu + 1; parts = (cu // 4) ** 2
            x0, y0 = int(jobs[i]["x"]) // 2, int(jobs[i]["y"]) // 2
            osyn = np.zeros(1, INTRA_SYN)
            for k in INTRA_SYN.names: osyn[0][k] = syn2[i][k]
            coder = snaps[i, :150].tobytes() + b"\0\0" + (int(snaps[i, 150]) | (int(snaps[i, 151]) << 8)).to_bytes(8, "little")
            arr_in = np.concatenate([res[i]["tr_idx"][None, :], res[i]["cbf"], res[i]["tskip"]]).reshape(-1)
            org = np.concatenate([C[k][y0:y0 + half, x0:x0 + half].reshape(-1) for k in range(2)]); win = np.concatenate([RC[k][y0 - 1:y0 - 1 + Wn, x0 - 1:x0 - 1 + Wn].reshape(-1) for k in range(2)])
            mode, dist, arr, ocoef, oreco, owin = oracle_intra_chroma_search(c, osyn[0], int(opts[i]["ts_fast"]), avs[i], org, win, arr_in, np.frombuffer(coder, np.uint8), cus[i])
            tag = (bd, i, cu, int(res[i]["tr_idx"][:parts].max()))
            assert int(cres[i]["best_mode"]) == mode and int(cres[i]["dist"]) == dist, (tag, cres[i], mode, dist)
            assert np.array_equal(res2[i]["cbf"][1:, :parts], arr[2:4, :parts]) and np.array_equal(res2[i]["tskip"][1:, :parts], arr[5:7, :parts]), tag
            assert np.array_equal(coef[o + cu * cu:o + cu * cu * 3 // 2], ocoef) and np.array_equal(reco[ro:ro + cu * cu // 2], oreco), tag
            for k in range(2): RC[k][y0 - 1:y0 - 1 + Wn, x0 - 1:x0 - 1 + Wn] = owin.reshape(2, Wn, Wn)[k]
            ts += int(arr[5:7, :parts].any()); modes.add(mode)
            o += cu * cu * 3 // 2; ro += cu * cu // 2
        assert np.array_equal(RC[0], R2[0]) and np.array_equal(RC[1], R2[1]), bd
        assert ts >= 3 and len(modes) >= 4, (bd, ts, modes)
        ctx.close()


def test_intra_cu_total_bits_encoder_calls(hp):
    """hop_intra_cu_total_bits (the counting part of xCheckRDCostIntra: header, luma and chroma directions, xEncodeTransform with the intra rules on the final levels, the RD
    cost) on the 40 calls recorded inside the encoder: bits, cost (the restatement's calcRdCost on the recorded distortion), all context states afterwards"""
    from goldutil import encoder_intracu_calls
    O = oracle(); O.hop_o_calc_rd_cost.restype = ctypes.c_double; O.hop_o_calc_rd_cost.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double]
    cases = list(encoder_intracu_calls())
    n = len(cases)
    jobs = np.zeros(n, hp.RQT_JOB_DTYPE); syn = np.zeros(n, hp.INTRA_CU_SYNTAX_DTYPE); res = np.zeros(n, hp.RQT_RESULT_DTYPE)
    snaps = np.zeros((n, hp.CABAC_CTX_BYTES), np.uint8); cus = np.zeros((n, hp.CABAC_CU_CTX_BYTES), np.uint8)
    for i, c in enumerate(cases):
        cfg = c["cfg"]; j = jobs[i]
        j["log2_cu"], j["ctx_index"], j["sign_hide"], j["use_ts"], j["log2_max_tu"], j["log2_min_tu_in_cu"] = int(cfg["log2_cu"]), i, cfg["sign_hide"], cfg["use_ts"], cfg["log2_max_tu"], cfg["log2_min_tu_in_cu"]
        j["lambda_rd"] = cfg["lambda_rd"]; j["lambda_rdoq"] = 1.0
        for k in ("part_nxn", "skip_flag", "skip_ctx", "is_min_cu", "luma_dir", "preds", "pred_num", "chroma_is_dm", "chroma_dir"): syn[i][k] = c["syn"][k]
        a = c["arr"].reshape(7, 256); res[i]["tr_idx"] = a[0]; res[i]["cbf"] = a[1:4]; res[i]["tskip"] = a[4:7]
        snaps[i, :150] = c["cin"]["ctx"]; left = int(c["cin"]["frac"]) & 32767; snaps[i, 150], snaps[i, 151] = left & 255, left >> 8
        cus[i] = c["cuin"]
    ctx = hp.Context(64, 64)
    dist = np.array([c["dist"] for c in cases], np.uint32)
    bits, cost, cx, cu = ctx.intra_cu_total_bits(jobs, syn, res, np.concatenate([c["coef"] for c in cases]), dist, snaps, cus)
    for i, c in enumerate(cases):
        assert int(bits[i]) == c["bits"], (i, int(c["cfg"]["log2_cu"]), int(bits[i]), c["bits"])
        assert float(cost[i]) == O.hop_o_calc_rd_cost(c["bits"], c["dist"], float(c["cfg"]["lambda_rd"])), i
        assert np.array_equal(cx[i, :150], c["cout"]["ctx"]) and (int(cx[i, 150]) | (int(cx[i, 151]) << 8)) == (int(c["cout"]["frac"]) & 32767), i
        assert np.array_equal(cu[i], c["cuout"]), i
    bad = res.copy(); bad[0]["tr_idx"][0] = 7
    with pytest.raises(hp.HopError):
        ctx.intra_cu_total_bits(jobs, syn, bad, np.concatenate([c["coef"] for c in cases]), dist, snaps, cus)
    ctx.close()


def test_intra_entries_empty_and_error_paths(hp):
    """the quadtree / search entries with nothing to do (n = 0 is not an error) and used wrongly (before the original was uploaded; a CU outside the picture; a snapshot
    index out of range): every failure is reported through the error code and hop_last_error, nothing is computed"""
    ctx = hp.Context(128, 128)
    jobs = np.zeros(1, hp.RQT_JOB_DTYPE); syn = np.zeros(1, hp.INTRA_CU_SYNTAX_DTYPE); opts = np.zeros(1, hp.INTRA_RQT_OPT_DTYPE); sj = np.zeros(1, hp.INTRA_SEARCH_JOB_DTYPE)
    res = np.zeros(1, hp.RQT_RESULT_DTYPE); snap = np.zeros((1, hp.CABAC_CTX_BYTES), np.uint8); cu = np.zeros((1, hp.CABAC_CU_CTX_BYTES), np.uint8)
    jobs["log2_cu"] = 4; jobs["qp_scaled"] = 30; jobs["log2_max_tu"] = 5; jobs["log2_min_tu_in_cu"] = 2; jobs["lambda_rd"] = 50.0; jobs["lambda_rdoq"] = 50.0; jobs["dist_weight"] = 1.0
    syn["pred_num"] = 3; syn["preds"] = (0, 1, 26); sj["sqrt_lambda"] = 7.0; sj["num_full_rd"] = 3; sj["left_dir"] = 1; sj["above_dir"] = 1
    with pytest.raises(hp.HopError):                                    # no original yet
        ctx.intra_rqt(jobs, syn, opts, snap, cu)
    with pytest.raises(hp.HopError):
        ctx.intra_luma_search(jobs, syn, opts, sj, snap, cu)
    with pytest.raises(hp.HopError):
        ctx.intra_chroma_search(jobs, syn, opts, res, snap, cu)
    ctx.upload_orig(np.full((128, 128), 128, np.int16), np.full((64, 64), 128, np.int16), np.full((64, 64), 128, np.int16))
    e = lambda a: a[:0]
    r0 = ctx.intra_rqt(e(jobs), e(syn), e(opts), snap, cu); assert len(r0[0]) == 0
    s0 = ctx.intra_luma_search(e(jobs), e(syn), e(opts), e(sj), snap, cu); assert len(s0[0]) == 0
    c0 = ctx.intra_chroma_search(e(jobs), e(syn), e(opts), e(res), snap, cu); assert len(c0[0]) == 0
    b0 = ctx.intra_cu_total_bits(e(jobs), e(syn), e(res), np.zeros(0, np.int32), np.zeros(0, np.uint32), snap, cu); assert len(b0[0]) == 0
    for field, value in (("x", 120), ("ctx_index", 1), ("log2_cu", 7), ("log2_min_tu_in_cu", 6)):
        bad = jobs.copy(); bad[field] = value
        with pytest.raises(hp.HopError):
            ctx.intra_rqt(bad, syn, opts, snap, cu)
        with pytest.raises(hp.HopError):
            ctx.intra_luma_search(bad, syn, opts, sj, snap, cu)
        with pytest.raises(hp.HopError):
            ctx.intra_chroma_search(bad, syn, opts, res, snap, cu)
    # and a legal call still works afterwards: a flat CU costs a few bits and reconstructs exactly
    opts["avail"] = 0; sj["rough_flags"] = 0
    sres, r, coef, reco = ctx.intra_luma_search(jobs, syn, opts, sj, snap, cu)
    assert int(sres[0]["dist"]) == 0 and not coef.any() and (reco == 128).all()
    ctx.close()


def test_inter_cu_skip_encoder_calls(hp):
    """hop_inter_cu_skip (encodeResAndCalcRdInterCU without residual) on the calls recorded inside the encoder, each CU placed in one picture with its recorded prediction
    and original: the three distortions, bits of skip flag + merge index, cost, context states, and the prediction copied into the reconstruction picture"""
    from goldutil import encoder_cuskip_calls
    cases = list(encoder_cuskip_calls())
    n = len(cases); G = 5; W = H = 64 * G
    assert n <= G * G
    O = [np.zeros((H, W), np.int16), np.zeros((H // 2, W // 2), np.int16), np.zeros((H // 2, W // 2), np.int16)]; P = [a.copy() for a in O]
    jobs = np.zeros(n, hp.RQT_JOB_DTYPE); syn = np.zeros(n, hp.CU_SYNTAX_DTYPE)
    snaps = np.zeros((n, hp.CABAC_CTX_BYTES), np.uint8); cus = np.zeros((n, hp.CABAC_CU_CTX_BYTES), np.uint8)
    for i, c in enumerate(cases):
        cfg = c["cfg"]; cu = 1 << int(cfg["log2_cu"]); n2 = cu * cu; x0, y0 = (i % G) * 64, (i // G) * 64
        for k, (a, b, w, xx, yy) in enumerate(((0, n2, cu, x0, y0), (n2, n2 + n2 // 4, cu // 2, x0 // 2, y0 // 2), (n2 + n2 // 4, n2 * 3 // 2, cu // 2, x0 // 2, y0 // 2))):
            O[k][yy:yy + w, xx:xx + w] = c["org"][a:b].reshape(w, w); P[k][yy:yy + w, xx:xx + w] = c["pred"][a:b].reshape(w, w)
        j = jobs[i]; j["x"], j["y"], j["log2_cu"], j["ctx_index"], j["lambda_rd"], j["dist_weight"] = x0, y0, int(cfg["log2_cu"]), i, cfg["lambda_rd"], cfg["dist_weight"][1:]
        j["log2_max_tu"] = 5; j["log2_min_tu_in_cu"] = 2; j["lambda_rdoq"] = 1.0
        syn[i]["skip_ctx"], syn[i]["max_merge_cand"], syn[i]["n_pu"] = c["nd"][0], c["nd"][2], 1; syn[i]["pu"][0]["merge_flag"] = 1; syn[i]["pu"][0]["merge_idx"] = c["nd"][1]
        snaps[i, :150] = c["cin"]["ctx"]; left = int(c["cin"]["frac"]) & 32767; snaps[i, 150], snaps[i, 151] = left & 255, left >> 8
        cus[i, :16] = c["cuin"]
    ctx = hp.Context(W, H)
    ctx.upload_orig(*O)
    for k in range(3):
        ctx.plane_upload("pred", k, P[k]); ctx.plane_upload("recon", k, np.full(P[k].shape, -7, np.int16))
    fin, bits, cost, cx, cu_out = ctx.inter_cu_skip(jobs, syn, snaps, cus)
    for i, c in enumerate(cases):
        assert [int(bits[i])] + [int(v) for v in fin[i, 1:]] == c["o4"] and int(fin[i, 0]) == 0 and float(cost[i]) == c["cost"], (i, bits[i], fin[i], c["o4"])
        assert np.array_equal(cx[i, :150], c["cout"]["ctx"]) and (int(cx[i, 150]) | (int(cx[i, 151]) << 8)) == (int(c["cout"]["frac"]) & 32767), i
        assert np.array_equal(cu_out[i, :16], c["cuout"]), i
    for k in range(3):
        R = ctx.recon_download(k)
        for i, c in enumerate(cases):
            cu = (1 << int(c["cfg"]["log2_cu"])) >> (1 if k else 0); xx, yy = ((i % G) * 64) >> (1 if k else 0), ((i // G) * 64) >> (1 if k else 0)
            assert np.array_equal(R[yy:yy + cu, xx:xx + cu], P[k][yy:yy + cu, xx:xx + cu]), (k, i)
    bad = syn.copy(); bad[0]["pu"][0]["merge_idx"] = 9
    with pytest.raises(hp.HopError):
        ctx.inter_cu_skip(jobs, bad, snaps, cus)
    ctx.close()


def test_intra_cu_device_classes_vs_staged_entries(hp, monkeypatch):
    """hop_intra_cu_device_classes (whole intra candidates device-resident: luma search -> chroma search -> distortion -> bits and cost, classes on separate streams) on the
    random CUs of the search tests: every output equals what the three host-array entries give stage by stage (those are checked against the restatement above), the
    syntax elements it hands back carry the decided directions and their most probable modes.
    The random availability tables of these cases are deliberately NOT causal (a node may be told that samples inside its own CU, not coded yet, are available: it then reads
    what the previous candidate pass left in the picture).  The walk that evaluates a PU's candidates side by side (k_iw_*: every candidate in a band of its own, started from
    the picture as it was before any candidate) equals the candidate-after-candidate forms only for causal tables -- what an encoder produces; that form is pinned by
    tests/test_gpu_spine.py against the reference encoder's decisions.  Here the walk runs its candidates one after the other (HOP_WALK_CAND=0)."""
    import torch
    monkeypatch.setenv("HOP_WALK_CAND", "0")
    bd = 8
    W, H, Y, R, jobs, syn, opts, cfgs, snaps, cus, avs = _irqt_random_cases(hp, hp.load(), bd)
    n = len(jobs); mid = 1 << (bd - 1)
    rng = np.random.default_rng(130)
    sj = np.zeros(n, hp.INTRA_SEARCH_JOB_DTYPE)
    sj["left_dir"] = rng.integers(0, 35, (n, 4)); sj["above_dir"] = rng.integers(0, 35, (n, 4)); sj["rough_flags"] = 1
    for i in range(n):
        sj[i]["sqrt_lambda"] = float(np.sqrt(jobs[i]["lambda_rd"])); sj[i]["num_full_rd"] = 8 if ((1 << int(jobs[i]["log2_cu"])) >> int(syn[i]["part_nxn"])) <= 8 else 3
    C = [np.clip(Y[::2, ::2] // 2 + 60 + k * 9, 0, 255).astype(np.int16) for k in range(2)]
    RC = [rng.integers(0, 256, (H // 2, W // 2)).astype(np.int16) for _ in range(2)]
    def fresh():
        ctx = hp.Context(W, H, bd)
        ctx.upload_orig(Y, C[0], C[1]); ctx.plane_upload("recon", 0, R)
        for k in range(2): ctx.plane_upload("recon", 1 + k, RC[k])
        return ctx
    # staged: the three host-array entries
    ctx = fresh()
    sres, res, coef_y, reco_y = ctx.intra_luma_search(jobs, syn, opts, sj, snaps, cus)
    syn2 = syn.copy(); syn2["luma_dir"] = sres["best_dir"]
    for i in range(n):                                                   # getIntraDirLumaPredictor per PU, as the search derived them
        nxn = int(syn[i]["part_nxn"])
        for pu in range(4 if nxn else 1):
            l = int(sres[i]["best_dir"][pu - 1]) if (nxn and pu & 1) else int(sj[i]["left_dir"][pu]); a = int(sres[i]["best_dir"][pu - 2]) if (nxn and pu & 2) else int(sj[i]["above_dir"][pu])
            syn2[i]["preds"][pu] = ((l, ((l + 29) % 32) + 2, ((l - 1) % 32) + 2) if l > 1 else (0, 1, 26)) if l == a else (l, a, 0 if (l and a) else (26 if l + a < 2 else 1))
            syn2[i]["pred_num"][pu] = 3
    cres, res2, coef_c, reco_c = ctx.intra_chroma_search(jobs, syn2, opts, res, snaps, cus)
    syn3 = syn2.copy(); syn3["chroma_is_dm"] = cres["best_mode"] == 36; syn3["chroma_dir"] = cres["best_mode"]
    coef = coef_y + coef_c
    dist = (sres["dist"] + cres["dist"]).astype(np.uint32)
    bits, cost, cx, cu_out = ctx.intra_cu_total_bits(jobs, syn3, res2, coef, dist, snaps, cus)
    rec_staged = [ctx.recon_download(k) for k in range(3)]
    ctx.close()
    # device-resident, classes on streams
    ctx = fresh()
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
    CLS = np.dtype([("n", "<i4"), ("part_nxn", "<i4"), ("num_full_rd", "<i4"), ("pad", "<i4"), ("cls", hp.RQT_JOB_DTYPE)] + [(k, "<u8") for k in
                   ("d_jobs", "d_syntax", "d_opts", "d_sjobs", "d_sresults", "d_results", "d_cresults", "d_coef", "d_reco_y", "d_reco_c", "d_syntax_out", "d_dist", "d_bits", "d_cost",
                    "d_ctx_out", "d_cu_ctx_out")])
    assert CLS.itemsize == 16 + hp.RQT_JOB_DTYPE.itemsize + 16 * 8
    groups = {}
    for i in range(n):
        j = jobs[i]; groups.setdefault((int(j["log2_cu"]), int(j["log2_min_tu_in_cu"]), int(j["sign_hide"]), int(j["use_ts"]), int(syn[i]["part_nxn"])), []).append(i)
    d_snap = up(snaps); d_cus = up(cus)
    descs = np.zeros(len(groups), CLS); keep = []
    for gi, (key, idx) in enumerate(sorted(groups.items())):
        m = len(idx); S = 1 << key[0]
        jj = jobs[idx].copy(); jj["ctx_index"] = idx
        bufs = dict(d_jobs=up(jj), d_syntax=up(syn[idx]), d_opts=up(opts[idx]), d_sjobs=up(sj[idx]),
                    d_sresults=torch.zeros(m * hp.INTRA_SEARCH_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev), d_results=torch.zeros(m * hp.RQT_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev),
                    d_cresults=torch.zeros(m * 8, dtype=torch.uint8, device=dev), d_coef=torch.zeros(m * S * S * 3 // 2, dtype=torch.int32, device=dev),
                    d_reco_y=torch.zeros(m * S * S, dtype=torch.int16, device=dev), d_reco_c=torch.zeros(m * S * S // 2, dtype=torch.int16, device=dev),
                    d_syntax_out=torch.zeros(m * hp.INTRA_CU_SYNTAX_DTYPE.itemsize, dtype=torch.uint8, device=dev), d_dist=torch.zeros(m, dtype=torch.int32, device=dev),
                    d_bits=torch.zeros(m, dtype=torch.int32, device=dev), d_cost=torch.zeros(m, dtype=torch.float64, device=dev),
                    d_ctx_out=torch.zeros(m * hp.CABAC_CTX_BYTES, dtype=torch.uint8, device=dev), d_cu_ctx_out=torch.zeros(m * hp.CABAC_CU_CTX_BYTES, dtype=torch.uint8, device=dev))
        d = descs[gi]; d["n"], d["part_nxn"], d["num_full_rd"], d["cls"] = m, key[4], int(sj[idx[0]]["num_full_rd"]), jj[0]
        for k2, t in bufs.items(): d[k2] = t.data_ptr()
        keep.append((idx, S, bufs))
    ctx.L.hop_intra_cu_device_classes.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    # four calls on the same buffers: every call starts from the same pictures and must leave the same results (work areas reused from call to call)
    for rep in range(4):
        ctx.plane_upload("recon", 0, R)
        for k in range(2): ctx.plane_upload("recon", 1 + k, RC[k])
        for idx, S, b in keep:
            for name in ("d_sresults", "d_results", "d_cresults", "d_coef", "d_reco_y", "d_reco_c", "d_syntax_out", "d_dist", "d_bits", "d_cost", "d_ctx_out", "d_cu_ctx_out"): b[name].zero_()
        ctx._chk(ctx.L.hop_intra_cu_device_classes(ctx.h, len(descs), descs.ctypes.data, d_snap.data_ptr(), d_cus.data_ptr()), "hop_intra_cu_device_classes")
        ctx.sync()
    coff = np.concatenate([[0], np.cumsum([(3 << (2 * int(j["log2_cu"]))) // 2 for j in jobs])]); yoff = np.concatenate([[0], np.cumsum([1 << (2 * int(j["log2_cu"])) for j in jobs])])
    for idx, S, b in keep:
        dn = lambda t, dt: np.frombuffer(t.cpu().numpy().tobytes(), dt)
        g_s = dn(b["d_sresults"], hp.INTRA_SEARCH_RESULT_DTYPE); g_r = dn(b["d_results"], hp.RQT_RESULT_DTYPE); g_c = dn(b["d_cresults"], np.dtype([("best_mode", "<i4"), ("dist", "<u4")]))
        g_y = dn(b["d_syntax_out"], hp.INTRA_CU_SYNTAX_DTYPE); g_co = b["d_coef"].cpu().numpy(); g_ry = b["d_reco_y"].cpu().numpy(); g_rc = b["d_reco_c"].cpu().numpy()
        g_b = b["d_bits"].cpu().numpy().view(np.uint32); g_k = b["d_cost"].cpu().numpy(); g_d = b["d_dist"].cpu().numpy().view(np.uint32)
        g_cx = b["d_ctx_out"].cpu().numpy().reshape(-1, hp.CABAC_CTX_BYTES); g_cu = b["d_cu_ctx_out"].cpu().numpy().reshape(-1, hp.CABAC_CU_CTX_BYTES)
        for t, i in enumerate(idx):
            parts = (S // 4) ** 2; npu = 4 if syn[i]["part_nxn"] else 1; tag = (i, S, npu)
            assert g_s[t].tobytes() == sres[i].tobytes() and int(g_c[t]["best_mode"]) == int(cres[i]["best_mode"]) and int(g_c[t]["dist"]) == int(cres[i]["dist"]), tag
            for name in ("tr_idx", "cbf", "tskip"): assert np.array_equal(g_r[t][name][..., :parts], res2[i][name][..., :parts]), (tag, name)
            assert np.array_equal(g_co[t * S * S * 3 // 2:(t + 1) * S * S * 3 // 2], coef[coff[i]:coff[i + 1]]), tag
            assert np.array_equal(g_ry[t * S * S:(t + 1) * S * S], reco_y[yoff[i]:yoff[i + 1]]) and np.array_equal(g_rc[t * S * S // 2:(t + 1) * S * S // 2], reco_c[yoff[i] // 2:yoff[i + 1] // 2]), tag
            assert int(g_d[t]) == int(dist[i]) and int(g_b[t]) == int(bits[i]) and float(g_k[t]) == float(cost[i]), (tag, g_b[t], bits[i])
            assert np.array_equal(g_cx[t], cx[i]) and np.array_equal(g_cu[t], cu_out[i]), tag
            for name in ("luma_dir", "preds", "pred_num"): assert np.array_equal(g_y[t][name][:npu], syn3[i][name][:npu]), (tag, name)
            assert int(g_y[t]["chroma_is_dm"]) == int(syn3[i]["chroma_is_dm"]) and int(g_y[t]["chroma_dir"]) == int(syn3[i]["chroma_dir"]), tag
    for k in range(3): assert np.array_equal(ctx.recon_download(k), rec_staged[k]), k
    assert len(groups) >= 8
    ctx.close()


def test_inter_cu_device_classes_vs_staged_entries(hp):
    """hop_inter_cu_device_classes (SS/GT candidates with residual, device-resident: quadtree -> root-cbf decision and reconstruction -> CU syntax bits -> cost, classes on
    separate streams; with the fixture's default one walk kernel per candidate, k_walk.inl) against the three host-array entries stage by stage (those are checked against the restatement and the recorded
    encoder calls above): 40 random CUs of all sizes, four calls (launch by launch twice, captured, replayed)"""
    import torch
    O = oracle(); O.hop_o_calc_rd_cost.restype = ctypes.c_double; O.hop_o_calc_rd_cost.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double]
    rng = np.random.default_rng(150)
    W, H, n = 512, 320, 40
    org = [np.full((H, W), 128, np.int16), np.full((H // 2, W // 2), 128, np.int16), np.full((H // 2, W // 2), 128, np.int16)]
    jobs = np.zeros(n, hp.RQT_JOB_DTYPE); syn = np.zeros(n, hp.CU_SYNTAX_DTYPE); snaps = np.zeros((n, hp.CABAC_CTX_BYTES), np.uint8); cus = np.zeros((n, hp.CABAC_CU_CTX_BYTES), np.uint8)
    L = hp.load(); L.hop_cabac_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]; L.hop_cabac_cu_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    for i in range(n):
        lg = 3 + (i % 4); cu = 1 << lg; x, y = 64 * (i % 8), 64 * (i // 8)
        yy, xx = np.mgrid[0:cu, 0:cu]; amp = float(rng.choice([4, 12, 40]))
        ry = amp * np.sin(xx * rng.uniform(0.1, 0.9) + rng.uniform(0, 3)) * np.cos(yy * rng.uniform(0.1, 0.9)) + rng.normal(0, amp / 4, (cu, cu))
        if rng.random() < 0.5:
            m = rng.random((cu, cu)) < 0.06; ry[m] += rng.choice([-1, 1], int(m.sum())) * 6 * amp
        org[0][y:y + cu, x:x + cu] += np.rint(ry).astype(np.int16)
        for k in (1, 2): org[k][y // 2:(y + cu) // 2, x // 2:(x + cu) // 2] += np.rint(ry[::2, ::2] * (0.5 if k == 1 else -0.4)).astype(np.int16)
        qp = int(rng.integers(22, 40)); lam = 0.57 * 2.0 ** ((qp - 12) / 3.0); w = float(rng.choice([1.0, 1.26]))
        j = jobs[i]; j["x"], j["y"], j["log2_cu"], j["qp_scaled"], j["ctx_index"] = x, y, lg, (qp, qp - 1, qp - 2), i
        j["sign_hide"], j["use_ts"], j["log2_max_tu"], j["log2_min_tu_in_cu"] = 1, 1, 5, max(2, lg - 2)
        j["lambda_rd"], j["lambda_rdoq"], j["dist_weight"] = lam, (lam, lam / w, lam / w), (w, w)
        s = syn[i]; s["n_pu"], s["max_merge_cand"], s["amp_acc"], s["is_min_cu"], s["skip_ctx"] = 1, 5, int(cu >= 16), int(cu == 8), int(rng.integers(0, 3))
        s["pu"][0]["mvd"] = rng.integers(-40, 41, 2); s["pu"][0]["gt_flag"] = int(rng.integers(0, 2)); s["pu"][0]["gt"] = rng.integers(-3, 4, 8)
        assert L.hop_cabac_init(snaps[i].ctypes.data, int(rng.integers(0, 5)), qp) == 0 and L.hop_cabac_cu_init(cus[i].ctypes.data, int(rng.integers(0, 5)), qp) == 0
        left = int(rng.integers(0, 32768)); snaps[i, 150], snaps[i, 151] = left & 255, left >> 8
    org = [np.clip(a, 0, 255) for a in org]
    def fresh():
        ctx = hp.Context(W, H)
        ctx.upload_orig(*org)
        for k in range(3):
            ctx.plane_upload("pred", k, np.full(org[k].shape, 128, np.int16)); ctx.plane_upload("recon", k, np.zeros(org[k].shape, np.int16))
        return ctx
    ctx = fresh()
    res, co, cx = ctx.rqt(jobs, snaps)
    res3, co3, fin3 = ctx.rqt_finish(jobs, res, co, cx)
    bits, skipped, bx, bu = ctx.inter_cu_bits(jobs, syn, res3, co3, snaps, cus)
    rec_staged = [ctx.recon_download(k) for k in range(3)]
    ctx.close()
    ctx = fresh()
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
    CLS = np.dtype([("n", "<i4"), ("pad", "<i4"), ("cls", hp.RQT_JOB_DTYPE)] + [(k, "<u8") for k in
                   ("d_jobs", "d_syntax", "d_results", "d_coef", "d_ctx_after", "d_finals", "d_bits", "d_skipped", "d_cost", "d_ctx_out", "d_cu_ctx_out")])
    groups = {}
    for i in range(n): groups.setdefault(int(jobs[i]["log2_cu"]), []).append(i)
    d_snap = up(snaps); d_cus = up(cus)
    descs = np.zeros(len(groups), CLS); keep = []
    for gi, (lg, idx) in enumerate(sorted(groups.items())):
        m = len(idx); S = 1 << lg; jj = jobs[idx].copy()
        b = dict(d_jobs=up(jj), d_syntax=up(syn[idx]), d_results=torch.zeros(m * hp.RQT_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev), d_coef=torch.zeros(m * S * S * 3 // 2, dtype=torch.int32, device=dev),
                 d_ctx_after=torch.zeros(m * hp.CABAC_CTX_BYTES, dtype=torch.uint8, device=dev), d_finals=torch.zeros(m * 4, dtype=torch.int32, device=dev), d_bits=torch.zeros(m, dtype=torch.int32, device=dev),
                 d_skipped=torch.zeros(m, dtype=torch.int32, device=dev), d_cost=torch.zeros(m, dtype=torch.float64, device=dev),
                 d_ctx_out=torch.zeros(m * hp.CABAC_CTX_BYTES, dtype=torch.uint8, device=dev), d_cu_ctx_out=torch.zeros(m * hp.CABAC_CU_CTX_BYTES, dtype=torch.uint8, device=dev))
        d = descs[gi]; d["n"], d["cls"] = m, jj[0]
        for k2, t in b.items(): d[k2] = t.data_ptr()
        keep.append((idx, S, b))
    ctx.L.hop_inter_cu_device_classes.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    for rep in range(4):
        for k in range(3): ctx.plane_upload("recon", k, np.zeros(org[k].shape, np.int16))
        for idx, S, b in keep:
            for name in ("d_results", "d_coef", "d_ctx_after", "d_finals", "d_bits", "d_skipped", "d_cost", "d_ctx_out", "d_cu_ctx_out"): b[name].zero_()
        ctx._chk(ctx.L.hop_inter_cu_device_classes(ctx.h, len(descs), descs.ctypes.data, d_snap.data_ptr(), d_cus.data_ptr()), "hop_inter_cu_device_classes")
        ctx.sync()
    coff = np.concatenate([[0], np.cumsum([(3 << (2 * int(j["log2_cu"]))) // 2 for j in jobs])])
    for idx, S, b in keep:
        g_r = np.frombuffer(b["d_results"].cpu().numpy().tobytes(), hp.RQT_RESULT_DTYPE); g_co = b["d_coef"].cpu().numpy(); g_f = b["d_finals"].cpu().numpy().view(np.uint32).reshape(-1, 4)
        g_b = b["d_bits"].cpu().numpy().view(np.uint32); g_s = b["d_skipped"].cpu().numpy().view(np.uint32); g_k = b["d_cost"].cpu().numpy()
        g_cx = b["d_ctx_out"].cpu().numpy().reshape(-1, hp.CABAC_CTX_BYTES); g_cu = b["d_cu_ctx_out"].cpu().numpy().reshape(-1, hp.CABAC_CU_CTX_BYTES)
        for t, i in enumerate(idx):
            parts = (S // 4) ** 2
            for name in ("tr_idx", "cbf", "tskip"): assert np.array_equal(g_r[t][name][..., :parts], res3[i][name][..., :parts]), (i, name)
            assert np.array_equal(g_co[t * S * S * 3 // 2:(t + 1) * S * S * 3 // 2], co3[coff[i]:coff[i + 1]]) and np.array_equal(g_f[t], fin3[i]), i
            assert int(g_b[t]) == int(bits[i]) and int(g_s[t]) == int(skipped[i]) and np.array_equal(g_cx[t], bx[i]) and np.array_equal(g_cu[t], bu[i]), i
            assert float(g_k[t]) == O.hop_o_calc_rd_cost(int(bits[i]), int(fin3[i][1]) + int(fin3[i][2]) + int(fin3[i][3]), float(jobs[i]["lambda_rd"])), i
    for k in range(3): assert np.array_equal(ctx.recon_download(k), rec_staged[k]), k
    # many distinct single-class calls (one lane forked, none of the extra streams): the prefixes of the first class, three rounds each -- seen, captured, replayed;
    # every call still gives the staged results for its CUs
    idx, S, b = keep[0]
    for rnd in range(3):
        for m in range(1, len(idx) + 1):
            one = descs[:1].copy(); one[0]["n"] = m
            for name in ("d_bits", "d_cost", "d_finals"): b[name].zero_()
            ctx._chk(ctx.L.hop_inter_cu_device_classes(ctx.h, 1, one.ctypes.data, d_snap.data_ptr(), d_cus.data_ptr()), "hop_inter_cu_device_classes")
            ctx.sync()
            g_b = b["d_bits"].cpu().numpy().view(np.uint32); g_f = b["d_finals"].cpu().numpy().view(np.uint32).reshape(-1, 4)
            assert [int(v) for v in g_b[:m]] == [int(bits[i]) for i in idx[:m]] and np.array_equal(g_f[:m], fin3[idx[:m]]) and not g_b[m:].any(), (rnd, m)
    ctx.close()
