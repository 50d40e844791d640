"""GPU parity tests proper: libhophip.so (through its C ABI) against the CPU oracle and against the
golden vectors produced by the reference's own code.  Bit-exact everywhere: the path is integer except
the GT warp, whose double arithmetic is reproduced in the reference's operation order (the final Pel
and every search cost must be identical)."""
import ctypes
import importlib.util
import os

import numpy as np
import pytest

from goldutil import crc, load, me_chain_scenarios
from hoputil import ROOT, Planes, lambda_for_qp, lenslet, oracle, p16

pytestmark = pytest.mark.gpu


def _hophip():
    spec = importlib.util.spec_from_file_location("hophip", os.path.join(ROOT, "hevc-hop_amd", "hophip.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def hp():
    return _hophip()


def _jobs_from_golden(hp, jobs, lc):
    a = np.zeros(len(jobs), hp.PU_JOB_DTYPE)
    for i, job in enumerate(jobs):
        (_, puX, puY, w, h, cuX, cuY, cuS, offx, offy, fr, fc, px, py, l, r, t, b, ox, oy, nA) = [int(v) for v in job[:21]]
        a[i]["pu_x"], a[i]["pu_y"], a[i]["w"], a[i]["h"] = puX, puY, w, h
        a[i]["rng_left"], a[i]["rng_right"], a[i]["rng_top"], a[i]["rng_bottom"] = l, r, t, b
        a[i]["off_x"], a[i]["off_y"], a[i]["pred_x"], a[i]["pred_y"] = ox, oy, px, py
        a[i]["lambda_cost"], a[i]["n_amvp"] = lc, nA
        a[i]["amvp"] = [int(v) for v in job[21:25]]
        a[i]["flags"] = hp.HOP_FLAG_FEN | hp.HOP_FLAG_HADME
    return a


def _res_row(r):
    """hop_pu_result -> the 27-entry layout of oracle/ref_harness.cpp:ref_me_pu"""
    return ([int(r["mv_int"][0]), int(r["mv_int"][1]), int(r["sad"]), int(r["not_valid"])] + [int(v) for v in r["half"]] +
            [int(v) for v in r["qter"]] + [int(r["frac_cost"]), int(r["gt_flag"])] + [int(v) for v in r["gt"]] +
            [int(r["cost"])] + [int(v) for v in r["mv_final"]] + [int(v) for v in r["half_final"]] + [int(v) for v in r["qter_final"]] +
            [int(r["mv_int"][0]), int(r["mv_int"][1])])


def test_library_identity(hp):
    L = hp.load()
    assert b"gfx950" in L.hop_version()


def test_ssref_commit_golden(hp):
    g = load("ssref_commit.npz")
    W, H = int(g["W"]), int(g["H"])
    Y, Cb, Cr = (g[k].astype(np.int16) for k in ("Y", "Cb", "Cr"))
    ctx = hp.Context(W, H)
    ctx.ssref_reset()
    assert (ctx.ssref_download(0) == -1).all() and (ctx.ssref_download(2) == -1).all()
    for (x, y, s), want in zip(g["order"], g["crcs"]):
        x, y, s = int(x), int(y), int(s)
        ctx.ssref_commit([(x, y, s)], Y[y:y + s, x:x + s], Cb[y // 2:(y + s) // 2, x // 2:(x + s) // 2], Cr[y // 2:(y + s) // 2, x // 2:(x + s) // 2])
        got = [crc(ctx.ssref_download(c)) for c in range(3)]
        assert got == [int(v) for v in want], (x, y, s)
    ctx.close()


def test_ssref_commit_batch_vs_oracle(hp):
    """several CUs per call, incl. all four picture corners, against the oracle's whole-picture re-extension"""
    O = oracle()
    W, H = 264, 200
    Y, Cb, Cr = lenslet(W, H, 15, 4)
    pl = Planes(W, H)
    ctx = hp.Context(W, H)
    batches = [[(0, 0, 64), (256, 0, 8), (0, 192, 8), (256, 192, 8)], [(64, 0, 64), (128, 64, 32), (160, 64, 16), (176, 80, 8)],
               [(192, 128, 64), (0, 64, 64), (256, 8, 8)]]
    for batch in batches:
        ys, cbs, crs = [], [], []
        for (x, y, s) in batch:
            ry = np.ascontiguousarray(Y[y:y + s, x:x + s]); rb = np.ascontiguousarray(Cb[y // 2:(y + s) // 2, x // 2:(x + s) // 2]); rr = np.ascontiguousarray(Cr[y // 2:(y + s) // 2, x // 2:(x + s) // 2])
            O.hop_o_ssref_commit_cu(pl.ptr00(0), pl.ptr00(1), pl.ptr00(2), W, H, x, y, s, p16(ry), p16(rb), p16(rr))
            ys.append(ry.ravel()); cbs.append(rb.ravel()); crs.append(rr.ravel())
        ctx.ssref_commit(batch, np.concatenate(ys), np.concatenate(cbs), np.concatenate(crs))
        assert np.array_equal(ctx.ssref_download(0), pl.bufY)
        assert np.array_equal(ctx.ssref_download(1), pl.bufCb)
        assert np.array_equal(ctx.ssref_download(2), pl.bufCr)
    ctx.close()


@pytest.mark.parametrize("stage", [1, 2, 3])
def test_me_chain_golden(hp, stage):
    """SS search / + fractional / + GT against the reference-generated golden vectors (84 PUs, 24 shapes)."""
    n = 0
    for pl, Y, jobs, outs, lc in me_chain_scenarios():
        g = load("me_chain.npz")
        ctx = hp.Context(pl.W, pl.H)
        ctx.upload_orig(Y, g["Cb"].astype(np.int16), g["Cr"].astype(np.int16))
        for c, b in enumerate((pl.bufY, pl.bufCb, pl.bufCr)):
            ctx.ssref_upload(c, b)
        res = ctx.me_search(_jobs_from_golden(hp, jobs, lc), stage)
        for job, r, want in zip(jobs, res, outs):
            got, want = _res_row(r), [int(v) for v in want]
            if want[3]:
                assert got[2:4] == want[2:4], (job, got, want)
                continue
            assert got[0:4] == want[0:4], (job, got, want)
            if stage >= 2:
                assert got[4:9] == want[4:9], (job, got, want)
            if stage >= 3:
                assert got[9:25] == want[9:25], (job, got, want)
            n += 1
        ctx.close()
    assert n == 80


def test_pred_inter_golden(hp):
    g = load("pred_inter.npz")
    Y, Cb, Cr = (g[k].astype(np.int16) for k in ("Y", "Cb", "Cr"))
    H, W = Y.shape
    ctx = hp.Context(W, H)
    ctx.upload_orig(Y, Cb, Cr)
    for c, k in enumerate(("bufY", "bufCb", "bufCr")):
        ctx.ssref_upload(c, g[k])
    jobs = []
    for job in g["jobs"]:
        j = hp.PredJob(*[int(v) for v in job[:7]])
        for k in range(8):
            j.gt[k] = int(job[7 + k])
        jobs.append(j)
    # the PUs of one call must not overlap (they share the prediction picture): one call per golden job
    off = 0
    for pj, job, n in zip(jobs, g["jobs"], g["out_len"]):
        oy, ocb, ocr = ctx.pred_inter([pj])
        want = g["out_flat"][off:off + n]
        got = np.concatenate([oy, ocb, ocr])
        assert np.array_equal(got, want), [int(v) for v in job]
        off += int(n)
    ctx.close()


def test_me_random_vs_oracle_10bit_and_sad(hp):
    """seeded random PUs against the oracle: 10-bit samples, HadamardME off (SAD costs), FEN off, empty and
    first-row/first-column windows, sentinel holes inside the window"""
    O = oracle()
    for bd, flags in ((10, hp.HOP_FLAG_FEN | hp.HOP_FLAG_HADME), (8, 0), (8, hp.HOP_FLAG_HADME), (10, hp.HOP_FLAG_FEN)):
        W, H = 256, 192
        Y, Cb, Cr = lenslet(W, H, 14, 31 + bd, bitdepth=bd)
        rng = np.random.default_rng(100 + bd + flags)
        pl = Planes(W, H)
        pl.y00()[:128, :W] = Y[:128]
        pl.y00()[128:192, :128] = Y[128:192, :128]
        pl.y00()[40:52, 60:90] = -1                      # a hole of sentinels inside the coded area
        by = np.ascontiguousarray(pl.y00()[0:8, 0:8]); z = np.zeros((4, 4), np.int16) - 1
        O.hop_o_ssref_commit_cu(pl.ptr00(0), pl.ptr00(1), pl.ptr00(2), W, H, 0, 0, 8, p16(by), p16(z), p16(z))
        lam, lc = lambda_for_qp(int(rng.integers(22, 38)))
        ctx = hp.Context(W, H, bit_depth=bd)
        ctx.upload_orig(Y, Cb, Cr)
        ctx.ssref_upload(0, pl.bufY)
        shapes = [(64, 64), (32, 32), (16, 16), (8, 8), (64, 32), (16, 32), (8, 4), (4, 8), (12, 16), (32, 24), (48, 64), (16, 4)]
        jobs = np.zeros(len(shapes) * 2, hp.PU_JOB_DTYPE)
        for i in range(len(jobs)):
            w, h = shapes[i % len(shapes)]
            cuS = 64 if max(w, h) > 32 else 32 if max(w, h) > 16 else 16 if max(w, h) > 8 else 8
            cuX, cuY = (128, 128) if i < len(shapes) else (int(rng.integers(0, 2)) * 64, 128)
            puX, puY = cuX + (cuS - w) * (i & 1), cuY + (cuS - h) * (i & 1)
            pred = (int(rng.integers(-50, 50)), int(rng.integers(-200, -30)))
            o6 = (ctypes.c_int * 6)()
            O.hop_o_set_search_range(W, H, cuX, cuY, cuS, 2 * 4 + cuX // 64, 4, pred[0], pred[1], 128, puX - cuX, puY - cuY, 0, int(cuX == 0), o6)
            j = jobs[i]
            j["pu_x"], j["pu_y"], j["w"], j["h"] = puX, puY, w, h
            j["rng_left"], j["rng_right"], j["rng_top"], j["rng_bottom"], j["off_x"], j["off_y"] = list(o6)
            j["pred_x"], j["pred_y"], j["lambda_cost"], j["n_amvp"] = pred[0], pred[1], lc, 2
            j["amvp"] = [pred[0], pred[1], int(rng.integers(-40, 40)), int(rng.integers(-300, -60))]
            j["flags"] = flags
        res = ctx.me_search(jobs, 3)
        for j, r in zip(jobs, res):
            out = (ctypes.c_int64 * 32)()
            org = np.ascontiguousarray(Y[j["pu_y"]:j["pu_y"] + j["h"], j["pu_x"]:j["pu_x"] + j["w"]])
            O.hop_o_me_pu(p16(org), int(j["w"]), pl.ptr00(0), pl.sy, int(j["pu_x"]), int(j["pu_y"]), int(j["w"]), int(j["h"]),
                          int(j["rng_left"]), int(j["rng_right"]), int(j["rng_top"]), int(j["rng_bottom"]), int(j["off_x"]), int(j["off_y"]),
                          int(j["pred_x"]), int(j["pred_y"]), 2, (ctypes.c_int * 4)(*[int(v) for v in j["amvp"]]), lc,
                          1 if flags & hp.HOP_FLAG_FEN else 0, 1 if flags & hp.HOP_FLAG_HADME else 0, bd, 3, out)
            want, got = list(out)[:27], _res_row(r)
            if want[3]:
                assert got[2:4] == want[2:4]
            else:
                assert got[:25] == want[:25], (bd, flags, [int(j[k]) for k in ("pu_x", "pu_y", "w", "h")], got, want)
        ctx.close()


def test_ss_families_vs_oracle(hp):
    """the five symmetric PUs of a CU handed over adjacently (the order of hop_enumerate_ctu_jobs) are searched in one
    shared pass: every member must still get exactly the single-PU result (oracle), with a common or with different
    predictors / windows per member, FEN on and off, 8 and 10 bit, sentinels in the window; and the developer switch
    HOP_SS_FAMILIES=0 (separate searches) must give the same bytes."""
    O = oracle()
    for bd, flags in ((8, hp.HOP_FLAG_FEN | hp.HOP_FLAG_HADME), (10, hp.HOP_FLAG_FEN), (8, 0)):
        W, H = 256, 192
        Y, Cb, Cr = lenslet(W, H, 14, 77 + bd, bitdepth=bd)
        rng = np.random.default_rng(500 + bd + flags)
        pl = Planes(W, H)
        pl.y00()[:128, :W] = Y[:128]
        pl.y00()[128:192, :128] = Y[128:192, :128]
        pl.y00()[70:84, 100:140] = -1                    # a hole of sentinels inside the coded area
        lam, lc = lambda_for_qp(int(rng.integers(22, 38)))
        cus = [(128, 128, 64), (64, 128, 64), (128, 128, 32), (160, 160, 32), (128, 128, 16), (176, 144, 16), (240, 176, 16),
               (128, 128, 8), (200, 136, 8), (0, 128, 8), (248, 184, 8), (128, 0, 16)]
        jobs = np.zeros(5 * len(cus), hp.PU_JOB_DTYPE)
        for c, (cuX, cuY, S) in enumerate(cus):
            h = S // 2
            common = (int(rng.integers(-50, 50)), int(rng.integers(-200, -30)))
            for m, (ox, oy, w, hh) in enumerate(((0, 0, S, S), (0, 0, h, S), (h, 0, h, S), (0, 0, S, h), (0, h, S, h))):
                pred = common if c % 2 == 0 else (int(rng.integers(-60, 60)), int(rng.integers(-220, -20)))
                o6 = (ctypes.c_int * 6)()
                O.hop_o_set_search_range(W, H, cuX, cuY, S, (cuY // 64) * 4 + cuX // 64, 4, pred[0], pred[1], 128, ox, oy,
                                         int(cuY == 0), int(cuX == 0), o6)
                j = jobs[5 * c + m]
                j["pu_x"], j["pu_y"], j["w"], j["h"] = cuX + ox, cuY + oy, w, hh
                j["rng_left"], j["rng_right"], j["rng_top"], j["rng_bottom"], j["off_x"], j["off_y"] = list(o6)
                j["pred_x"], j["pred_y"], j["lambda_cost"], j["n_amvp"] = pred[0], pred[1], lc, 0
                j["flags"] = flags
        outs = []
        for fam in ("1", "0"):
            os.environ["HOP_SS_FAMILIES"] = fam
            try:
                ctx = hp.Context(W, H, bit_depth=bd)
            finally:
                del os.environ["HOP_SS_FAMILIES"]
            ctx.upload_orig(Y, Cb, Cr)
            ctx.ssref_upload(0, pl.bufY)
            outs.append(ctx.me_search(jobs, 1).copy())
            ctx.close()
        assert outs[0].tobytes() == outs[1].tobytes()
        for j, r in zip(jobs, outs[0]):
            out = (ctypes.c_int64 * 32)()
            org = np.ascontiguousarray(Y[j["pu_y"]:j["pu_y"] + j["h"], j["pu_x"]:j["pu_x"] + j["w"]])
            O.hop_o_me_pu(p16(org), int(j["w"]), pl.ptr00(0), pl.sy, int(j["pu_x"]), int(j["pu_y"]), int(j["w"]), int(j["h"]),
                          int(j["rng_left"]), int(j["rng_right"]), int(j["rng_top"]), int(j["rng_bottom"]), int(j["off_x"]), int(j["off_y"]),
                          int(j["pred_x"]), int(j["pred_y"]), 0, (ctypes.c_int * 4)(0, 0, 0, 0), lc,
                          1 if flags & hp.HOP_FLAG_FEN else 0, 1 if flags & hp.HOP_FLAG_HADME else 0, bd, 1, out)
            want, got = list(out)[:27], _res_row(r)
            if want[3]:
                assert got[2:4] == want[2:4], (bd, flags, [int(j[k]) for k in ("pu_x", "pu_y", "w", "h")], got, want)
            else:
                assert got[:4] == want[:4], (bd, flags, [int(j[k]) for k in ("pu_x", "pu_y", "w", "h")], got, want)


def test_ss_family_cells_vs_oracle(hp):
    """whole-CTU enumerations (425 PUs each, hop_enumerate_ctu_jobs): the CUs of equal size inside a 64-column cell of the
    picture share one staged window; one CTU with a common predictor, one with a different predictor per PU, next to the
    uncoded (sentinel) area.  Every PU against the oracle, and the developer switch must not change a byte."""
    O = oracle()
    W, H, bd = 256, 192, 8
    flags = hp.HOP_FLAG_FEN | hp.HOP_FLAG_HADME
    Y, Cb, Cr = lenslet(W, H, 14, 91, bitdepth=bd)
    rng = np.random.default_rng(4242)
    pl = Planes(W, H)
    pl.y00()[:128, :W] = Y[:128]
    pl.y00()[128:192, :128] = Y[128:192, :128]
    lam, lc = lambda_for_qp(30)
    L = hp.load()
    parts = []
    for ctu, vary in ((2 * 4 + 2, False), (1 * 4 + 3, True)):
        buf = np.zeros(425, hp.PU_JOB_DTYPE)
        n = L.hop_enumerate_ctu_jobs(W, H, ctu, 128, (ctypes.c_int * 2)(4, -72), 0, None, lc, flags, 0, buf.ctypes.data, None, 425)
        assert n == 425
        if vary:
            cuX, cuY = (ctu % 4) * 64, (ctu // 4) * 64
            for j in buf:
                pred = (int(rng.integers(-40, 40)), int(rng.integers(-120, -40)))
                w, h = int(j["w"]), int(j["h"]); S = max(w, h)
                cx, cy = int(j["pu_x"]) // S * S, int(j["pu_y"]) // S * S
                o6 = (ctypes.c_int * 6)()
                O.hop_o_set_search_range(W, H, cx, cy, S, ctu, 4, pred[0], pred[1], 128, int(j["pu_x"]) - cx, int(j["pu_y"]) - cy,
                                         int(cy == 0), int(cx == 0), o6)
                j["rng_left"], j["rng_right"], j["rng_top"], j["rng_bottom"], j["off_x"], j["off_y"] = list(o6)
                j["pred_x"], j["pred_y"] = pred
        parts.append(buf)
    jobs = np.concatenate(parts)
    outs = []
    for fam in ("1", "0"):
        os.environ["HOP_SS_FAMILIES"] = fam
        try:
            ctx = hp.Context(W, H, bit_depth=bd)
        finally:
            del os.environ["HOP_SS_FAMILIES"]
        ctx.upload_orig(Y, Cb, Cr)
        ctx.ssref_upload(0, pl.bufY)
        outs.append(ctx.me_search(jobs, 1).copy())
        ctx.close()
    assert outs[0].tobytes() == outs[1].tobytes()
    for j, r in zip(jobs, outs[0]):
        out = (ctypes.c_int64 * 32)()
        org = np.ascontiguousarray(Y[j["pu_y"]:j["pu_y"] + j["h"], j["pu_x"]:j["pu_x"] + j["w"]])
        O.hop_o_me_pu(p16(org), int(j["w"]), pl.ptr00(0), pl.sy, int(j["pu_x"]), int(j["pu_y"]), int(j["w"]), int(j["h"]),
                      int(j["rng_left"]), int(j["rng_right"]), int(j["rng_top"]), int(j["rng_bottom"]), int(j["off_x"]), int(j["off_y"]),
                      int(j["pred_x"]), int(j["pred_y"]), 0, (ctypes.c_int * 4)(0, 0, 0, 0), lc, 1, 1, bd, 1, out)
        want, got = list(out)[:27], _res_row(r)
        if want[3]:
            assert got[2:4] == want[2:4], ([int(j[k]) for k in ("pu_x", "pu_y", "w", "h")], got, want)
        else:
            assert got[:4] == want[:4], ([int(j[k]) for k in ("pu_x", "pu_y", "w", "h")], got, want)


def test_batch_split_is_invisible(hp):
    """hop_me_search_device cuts large batches into parts that run on separate streams (HOP_LANES, default 2); one lane,
    two lanes and three lanes must give the same bytes for the whole chain (SS + frac + GT)."""
    W, H = 512, 384
    Y, Cb, Cr = lenslet(W, H, 15, 7)
    lam, lc = lambda_for_qp(32)
    L = hp.load()
    wctu, hctu = W // 64, H // 64
    jobs = np.zeros(425 * wctu * hctu, hp.PU_JOB_DTYPE)
    n = 0
    for a in range(wctu * hctu):
        n += L.hop_enumerate_ctu_jobs(W, H, a, 128, (ctypes.c_int * 2)(0, -60), 2, (ctypes.c_int * 4)(0, -60, -60, 0), lc,
                                      hp.HOP_FLAG_FEN | hp.HOP_FLAG_HADME, 0, jobs.ctypes.data + n * jobs.itemsize, None, len(jobs) - n)
    assert n == len(jobs) >= 16384
    rects = np.array([[x, y, 64] for y in range(0, H, 64) for x in range(0, W, 64)], np.int32)
    outs = []
    for lanes in ("1", "2", "3"):
        os.environ["HOP_LANES"] = lanes
        try:
            ctx = hp.Context(W, H)
        finally:
            del os.environ["HOP_LANES"]
        ctx.upload_orig(Y, Cb, Cr)
        ctx.ssref_reset()
        ctx.ssref_commit(rects, Y, Cb, Cr)                  # a fully reconstructed reference
        outs.append(ctx.me_search(jobs, 3).copy())
        ctx.close()
    assert int(np.sum(outs[0]["gt_flag"])) > 0
    assert outs[0].tobytes() == outs[1].tobytes() == outs[2].tobytes()


def test_distortion_vs_oracle(hp):
    O = oracle()
    W, H = 128, 128
    Y, Cb, Cr = lenslet(W, H, 15, 77)
    ctx = hp.Context(W, H)
    ctx.upload_orig(Y, Cb, Cr)
    # prediction picture := plain copy of a shifted reference through hop_pred_inter
    pl = Planes(W, H)
    pl.y00()[:H, :W] = np.roll(Y, 3, 1); pl.bufCb[40:40 + H // 2, 40:40 + W // 2] = np.roll(Cb, 1, 0); pl.bufCr[40:40 + H // 2, 40:40 + W // 2] = Cr[::-1]
    for c, b in enumerate((pl.bufY, pl.bufCb, pl.bufCr)):
        ctx.ssref_upload(c, b)
    jobs = [hp.PredJob(x, y, 64, 64, 0, 0, 0) for y in (0, 64) for x in (0, 64)]
    ctx.pred_inter(jobs)
    P = [ctx.pred_download(c) for c in range(3)]
    assert np.array_equal(P[0], pl.y00()[:H, :W])
    dj, want = [], []
    for (x, y, w, h) in ((0, 0, 64, 64), (64, 32, 32, 16), (8, 8, 8, 8), (16, 4, 16, 4), (32, 64, 12, 16), (4, 4, 8, 4), (0, 64, 64, 48)):
        for comp in range(3):
            o_, p_ = (Y, Cb, Cr)[comp], P[comp]
            ww, hh, xx, yy = (w, h, x, y) if comp == 0 else (w // 2, h // 2, x // 2, y // 2)
            a = np.ascontiguousarray(o_[yy:yy + hh, xx:xx + ww]); b = np.ascontiguousarray(p_[yy:yy + hh, xx:xx + ww])
            for kind, fn in ((hp.HOP_DIST_SAD, lambda: O.hop_o_sad(p16(a), ww, p16(b), ww, ww, hh, 8, 0)),
                             (hp.HOP_DIST_SSE, lambda: O.hop_o_sse(p16(a), ww, p16(b), ww, ww, hh, 8)),
                             (hp.HOP_DIST_HADS, lambda: O.hop_o_hads(p16(a), ww, p16(b), ww, ww, hh, 8))):
                if kind == hp.HOP_DIST_HADS and (ww % 4 or hh % 4):
                    continue
                if (ww % 4 or hh % 4):
                    continue
                dj.append(hp.DistJob(x, y, w, h, comp, kind)); want.append(fn())
    got = ctx.distortion(dj)
    assert [int(v) for v in got] == [int(v) for v in want]
    ctx.close()


def test_errors_are_reported(hp):
    with pytest.raises(hp.HopError):
        hp.Context(100, 64)                       # not a multiple of 8
    ctx = hp.Context(64, 64)
    jobs = np.zeros(1, hp.PU_JOB_DTYPE)
    jobs[0]["w"], jobs[0]["h"] = 64, 64
    with pytest.raises(hp.HopError):              # search before the original was uploaded
        ctx.me_search(jobs, 1)
    ctx.upload_orig(np.zeros((64, 64), np.int16), np.zeros((32, 32), np.int16), np.zeros((32, 32), np.int16))
    jobs[0]["w"] = 20                             # not a legal PU width
    with pytest.raises(hp.HopError):
        ctx.me_search(jobs, 1)
    ctx.close()


def test_encoder_calls_replay(hp):
    """hop_me_search on the calls of a real encode (tests/golden/encoder_calls.npz: 95 PUs of 20 shapes incl. AMP, sampled from the shim
    encoder whose bitstream equals the unmodified reference's): each PU with the SS-reference state, predictor, AMVP list, range and
    causality offsets the reference's xMotionEstimation handed to its three search members, against what those members returned."""
    from goldutil import encoder_calls
    ctx = None
    n = 0
    for pl, Y, m in encoder_calls():
        if ctx is None:
            ctx = hp.Context(pl.W, pl.H)
            c = np.full((pl.H // 2, pl.W // 2), 128, np.int16)
            ctx.upload_orig(Y, c, c)
            ctx.ssref_upload(1, pl.bufCb); ctx.ssref_upload(2, pl.bufCr)
        ctx.ssref_upload(0, pl.bufY)
        a = np.zeros(1, hp.PU_JOB_DTYPE)
        a[0]["pu_x"], a[0]["pu_y"], a[0]["w"], a[0]["h"] = m[0:4]
        a[0]["rng_left"], a[0]["rng_right"], a[0]["rng_top"], a[0]["rng_bottom"], a[0]["off_x"], a[0]["off_y"], a[0]["pred_x"], a[0]["pred_y"] = m[4:12]
        a[0]["lambda_cost"], a[0]["n_amvp"], a[0]["amvp"] = m[12], m[15], m[16:20]
        a[0]["flags"] = (hp.HOP_FLAG_FEN if m[13] else 0) | (hp.HOP_FLAG_HADME if m[14] else 0)
        got = _res_row(ctx.me_search(a, 3)[0])
        assert got[0:4] == m[20:23] + [0], (m[:20], got[0:4], m[20:23])
        assert got[4:9] == m[23:28], (m[:20], got[4:9], m[23:28])
        assert got[9:25] == m[28:44], (m[:20], got[9:25], m[28:44])
        n += 1
    ctx.close()
    assert n >= 90


def _warp_jobs(hp, O, rng, W, H, lc, n):
    """n random PUs of every PU shape of the RD tree (incl. AMP) inside the coded part of the plane, random predictors"""
    shapes = [(64, 64), (32, 32), (16, 16), (8, 8), (64, 32), (32, 64), (32, 16), (16, 32), (16, 8), (8, 16), (8, 4), (4, 8), (12, 16), (16, 12), (4, 16), (16, 4),
              (24, 32), (32, 24), (8, 32), (32, 8), (48, 64), (64, 48), (16, 64), (64, 16)]
    jobs = np.zeros(n, hp.PU_JOB_DTYPE)
    for i in range(n):
        w, h = shapes[i % len(shapes)]
        cuS = 64 if max(w, h) > 32 else 32 if max(w, h) > 16 else 16 if max(w, h) > 8 else 8
        cuX, cuY = 64 * int(rng.integers(1, 3)), 128
        puX, puY = cuX + (cuS - w) * int(rng.integers(0, 2)), cuY + (cuS - h) * int(rng.integers(0, 2))
        pred = (int(rng.integers(-120, 120)), int(rng.integers(-260, -40)))
        o6 = (ctypes.c_int * 6)()
        O.hop_o_set_search_range(W, H, cuX, cuY, cuS, 2 * 4 + cuX // 64, 4, pred[0], pred[1], 128, puX - cuX, puY - cuY, 0, 0, o6)
        j = jobs[i]
        j["pu_x"], j["pu_y"], j["w"], j["h"] = puX, puY, w, h
        j["rng_left"], j["rng_right"], j["rng_top"], j["rng_bottom"], j["off_x"], j["off_y"] = list(o6)
        j["pred_x"], j["pred_y"], j["lambda_cost"], j["n_amvp"] = pred[0], pred[1], lc, 2
        j["amvp"] = [pred[0], pred[1], int(rng.integers(-80, 80)), int(rng.integers(-300, -60))]
        j["flags"] = hp.HOP_FLAG_FEN | hp.HOP_FLAG_HADME
    return jobs


def test_integer_warp_vs_double_warp_100k_candidates(hp):
    """The GT search kernel evaluates its candidate warps in exact integer arithmetic (k_gt_search.hip: a proof of equivalence, not the reference's doubles); the
    restatement evaluates them as the reference does (double homography, x86 truncation).  Seeded random PUs of all 24 PU shapes, random predictors, on lenslet content
    with a sentinel hole: every field of every result must be equal -- a single sample of a single candidate warp rounding the other way changes a cost -- over at least
    1e5 candidate warps (counted by the restatement)."""
    O = oracle()
    O.hop_o_warp_counter.restype = ctypes.c_long
    W, H = 256, 192
    total, pus = 0, 0
    for seed in range(3):
        Y, Cb, Cr = lenslet(W, H, 13 + seed, 500 + seed)
        rng = np.random.default_rng(900 + seed)
        pl = Planes(W, H)
        pl.y00()[:128, :W] = Y[:128]
        pl.y00()[128:192, :64] = Y[128:192, :64]
        pl.y00()[70:80, 100:140] = -1
        by = np.ascontiguousarray(pl.y00()[0:8, 0:8]); z = np.zeros((4, 4), np.int16) - 1
        O.hop_o_ssref_commit_cu(pl.ptr00(0), pl.ptr00(1), pl.ptr00(2), W, H, 0, 0, 8, p16(by), p16(z), p16(z))
        lam, lc = lambda_for_qp(int(rng.integers(24, 40)))
        ctx = hp.Context(W, H)
        ctx.upload_orig(Y, Cb, Cr)
        ctx.ssref_upload(0, pl.bufY)
        jobs = _warp_jobs(hp, O, rng, W, H, lc, 240)
        res = ctx.me_search(jobs, 3)
        O.hop_o_warp_counter(1)
        for j, r in zip(jobs, res):
            out = (ctypes.c_int64 * 32)()
            org = np.ascontiguousarray(Y[j["pu_y"]:j["pu_y"] + j["h"], j["pu_x"]:j["pu_x"] + j["w"]])
            O.hop_o_me_pu(p16(org), int(j["w"]), pl.ptr00(0), pl.sy, int(j["pu_x"]), int(j["pu_y"]), int(j["w"]), int(j["h"]),
                          int(j["rng_left"]), int(j["rng_right"]), int(j["rng_top"]), int(j["rng_bottom"]), int(j["off_x"]), int(j["off_y"]),
                          int(j["pred_x"]), int(j["pred_y"]), 2, (ctypes.c_int * 4)(*[int(v) for v in j["amvp"]]), lc, 1, 1, 8, 3, out)
            want, got = list(out)[:27], _res_row(r)
            if want[3]:
                assert got[2:4] == want[2:4]
            else:
                assert got[:25] == want[:25], (seed, [int(j[k]) for k in ("pu_x", "pu_y", "w", "h")], got, want)
                pus += 1
        total += O.hop_o_warp_counter(1)
        ctx.close()
    print("GT searches compared: %d PUs, %d candidate warps" % (pus, total))
    assert total >= 100000, total
