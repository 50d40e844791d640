"""GPU: the SAO encoder of the library -- hop_sao_stats, hop_sao_frame (statistics kernel, host decision, offsetting kernel), hop_sao_apply -- through the C ABI.

1. the fixtures of the reference's own TEncSampleAdaptiveOffset::SAOProcess (tests/golden/sao_ref.npz): the statistics equal the reference's m_statData, the coded
   parameters equal the ones the reference wrote into the picture, every sample of the three output planes equals the reference's;
2. two pictures in one stacked context, each treated as a picture of its own;
3. at the full frame size (7680 x 5376 tiled from a fixture: 10 080 CTUs) statistics and offsetting against the CPU restatement."""
import ctypes
import math
import os
import sys
import time

import numpy as np
import pytest

from hoputil import ROOT, SAO_PARAM_DTYPE, oracle, same_coded, sao_cases

pytestmark = pytest.mark.gpu


def _hp():
    sys.path.insert(0, os.path.join(ROOT, "hevc-hop_amd"))
    import hophip
    return hophip


def _ptrs(ps):
    a = [np.ascontiguousarray(p, np.int16) for p in ps]
    return a, (ctypes.c_void_p * 3)(*[p.ctypes.data for p in a])


@pytest.mark.parametrize("case", sao_cases(), ids=lambda c: "%s_%dx%d_%dbit" % (c["key"], c["W"], c["H"], c["bd"]))
def test_sao_equals_the_reference_encoder(case):
    hp = _hp()
    ctx = hp.Context(case["W"], case["H"], case["bd"])
    ctx.upload_orig(*case["org"])
    for c in range(3):
        ctx.plane_upload("recon", c, case["in"][c])
    stats = ctx.sao_stats()
    assert np.array_equal(stats, case["stats"]), np.argwhere(stats != case["stats"])[:5]
    coded = ctx.sao_frame(case["lambda"], case["slice_type"], case["qp"], case["rd_fraction"])
    assert same_coded(coded, case["coded"])
    for c in range(3):
        got = ctx.recon_download(c)
        assert np.array_equal(got, case["out"][c]), (c, np.argwhere(got != case["out"][c])[:5])
    ctx.close()


def test_sao_stacked_pictures():
    hp = _hp()
    a, b = [c for c in sao_cases() if (c["W"], c["H"]) == (200, 104)][:2]
    ctx = hp.Context(200, 104, pictures=2)
    ctx.upload_orig(ctx.stack([a["org"][0], b["org"][0]]), ctx.stack([a["org"][1], b["org"][1]], True), ctx.stack([a["org"][2], b["org"][2]], True))
    ctx.plane_upload("recon", 0, ctx.stack([a["in"][0], b["in"][0]]))
    for c in (1, 2):
        ctx.plane_upload("recon", c, ctx.stack([a["in"][c], b["in"][c]], True))
    stats = ctx.sao_stats()
    assert np.array_equal(stats[:a["n"]], a["stats"]) and np.array_equal(stats[a["n"]:], b["stats"])
    coded = ctx.sao_frame(a["lambda"], a["slice_type"], a["qp"], a["rd_fraction"])           # both with the first fixture's lambdas: the first picture is the fixture
    assert same_coded(coded[:a["n"]], a["coded"])
    for c in range(3):
        assert np.array_equal(ctx.unstack(ctx.recon_download(c), c > 0)[0], a["out"][c]), c
    ctx.close()


def test_sao_full_size_against_the_restatement():
    hp = _hp(); O = oracle()
    base = [c for c in sao_cases() if (c["W"], c["H"]) == (256, 192)][0]
    nx, ny = 30, 28
    W, H = 256 * nx, 192 * ny
    org = [np.ascontiguousarray(np.tile(p, (ny, nx))) for p in base["org"]]; src = [np.ascontiguousarray(np.tile(p, (ny, nx))) for p in base["in"]]
    n = (W // 64) * (H // 64)
    ctx = hp.Context(W, H)
    ctx.upload_orig(*org)
    for c in range(3):
        ctx.plane_upload("recon", c, src[c])
    t0 = time.time(); stats = ctx.sao_stats(); t1 = time.time()
    _, psrc = _ptrs(src); _, porg = _ptrs(org)
    want = np.zeros((n, 3, 5, 32, 2), np.int32)
    assert O.hop_o_sao_stats(W, H, 8, psrc, porg, want.ctypes.data_as(ctypes.c_void_p)) == 0
    assert np.array_equal(stats, want)
    t2 = time.time(); coded = ctx.sao_frame(base["lambda"], base["slice_type"], base["qp"], base["rd_fraction"]); t3 = time.time()
    # the decision again on the host (the library's own entry), then the restated offsetting with its reconstructed parameters
    L = ctx.L
    p = hp.SaoParams((ctypes.c_double * 3)(*base["lambda"]), (ctypes.c_int32 * 3)(1, 1, 1), base["slice_type"], base["qp"], base["rd_fraction"])
    c2 = np.zeros((n, 3), SAO_PARAM_DTYPE); recon = np.zeros((n, 3), SAO_PARAM_DTYPE)
    L.hop_sao_decide.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4
    assert L.hop_sao_decide(n, W // 64, 8, want.ctypes.data, ctypes.addressof(p), c2.ctypes.data, recon.ctypes.data) == 0
    assert coded.tobytes() == c2.tobytes()
    out = [np.zeros_like(a) for a in src]; pout = (ctypes.c_void_p * 3)(*[a.ctypes.data for a in out])
    assert O.hop_o_sao_apply(W, H, 8, psrc, recon.ctypes.data_as(ctypes.c_void_p), pout) == 0
    for c in range(3):
        assert np.array_equal(ctx.recon_download(c), out[c]), c
    modes = np.bincount(coded["mode"].reshape(-1).astype(np.int64), minlength=3)
    print("SAO %dx%d: statistics %.1f ms (with download), whole hop_sao_frame %.1f ms; off / new / merge %s" % (W, H, (t1 - t0) * 1e3, (t3 - t2) * 1e3, modes.tolist()))
    ctx.close()


def test_psnr_against_numpy():
    """hop_psnr: the sums of squared differences and the reference's formula (TEncGOP.cpp:2449-2456), two stacked pictures, one of them exact"""
    hp = _hp()
    a = sao_cases()[0]
    W, H = a["W"], a["H"]
    ctx = hp.Context(W, H, pictures=2)
    ctx.upload_orig(ctx.stack([a["org"][0], a["org"][0]]), ctx.stack([a["org"][1], a["org"][1]], True), ctx.stack([a["org"][2], a["org"][2]], True))
    ctx.plane_upload("recon", 0, ctx.stack([a["in"][0], a["org"][0]]))
    for c in (1, 2):
        ctx.plane_upload("recon", c, ctx.stack([a["in"][c], a["org"][c]], True))
    ssd, ps = ctx.psnr()
    for c in range(3):
        want = int(((a["org"][c].astype(np.int64) - a["in"][c].astype(np.int64)) ** 2).sum())
        assert int(ssd[0, c]) == want and int(ssd[1, c]) == 0
        ref = 255.0 * 255.0 * W * H / (4.0 if c else 1.0)
        assert abs(ps[0, c] - 10.0 * math.log10(ref / want)) <= 1e-12 * ps[0, c] and ps[1, c] == 99.99          # the sums are exact; the dB value to 1e-12 relative (log10 of two libraries)
    ctx.close()
