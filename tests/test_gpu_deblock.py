"""GPU: hop_deblock_frame (csrc/k_deblock.hip) -- the deblocking filter over the resident reconstruction, SURVEY 8(f)-3 -- through the C ABI.

1. the fixtures of the reference's own TComLoopFilter::loopFilterPic (tests/golden/deblock_ref.npz): every sample of the three planes equals the reference's;
2. the disable flag; 3. two pictures in one stacked context, each filtered as a picture of its own; 4. at the full 7728x5368-class size (a 7680x5376 picture tiled from a
fixture: 10 080 CTUs, with new edges between unrelated CUs at every seam) against the CPU restatement; 5. after a real hop_encode_frame (a picture with partial CTUs)
against the restatement fed with the same partition data."""
import os
import sys
import time

import numpy as np
import pytest

from hoputil import ROOT, deblock_cases, lenslet, oracle_deblock, tile_deblock_case

pytestmark = pytest.mark.gpu


def _hp():
    sys.path.insert(0, os.path.join(ROOT, "hevc-hop_amd"))
    import hophip
    return hophip


def _run(hp, W, H, params, parts, planes, disable=0, pictures=1, bd=8):
    ctx = hp.Context(W, H, bd, pictures=pictures)
    for c in range(3):
        ctx.plane_upload("recon", c, planes[c])
    t0 = time.time()
    ctx.deblock_frame(parts, params[0], params[1], params[2], params[3], params[4], disable)
    dt = time.time() - t0
    out = [ctx.recon_download(c) for c in range(3)]
    ctx.close()
    return out, dt


@pytest.mark.parametrize("case", deblock_cases(), ids=lambda c: "%s_%dx%d_qp%d_%dbit" % (c[0], c[1], c[2], c[3][0], c[7]))
def test_deblock_equals_the_reference_filter(case):
    hp = _hp()
    key, W, H, params, parts, pin, pout, bd = case
    got, _ = _run(hp, W, H, params, parts, pin, bd=bd)
    for c in range(3):
        assert np.array_equal(got[c], pout[c]), (key, c, np.argwhere(got[c] != pout[c])[:5])
    off, _ = _run(hp, W, H, params, parts, pin, disable=1, bd=bd)
    assert all(np.array_equal(a, b) for a, b in zip(off, pin))


def test_deblock_stacked_pictures():
    hp = _hp()
    cases = [c for c in deblock_cases() if (c[1], c[2]) == (200, 104)][:2]
    (_, W, H, params, parts0, pin0, pout0, _), (_, _, _, _, parts1, pin1, _, _) = cases
    want1 = oracle_deblock(W, H, params, parts1, pin1)                     # the second fixture with the first one's parameters
    ctx = hp.Context(W, H, pictures=2)
    ctx.plane_upload("recon", 0, ctx.stack([pin0[0], pin1[0]]))
    for c in (1, 2):
        ctx.plane_upload("recon", c, ctx.stack([pin0[c], pin1[c]], True))
    ctx.deblock_frame(np.concatenate([parts0, parts1]), *params)
    for c in range(3):
        got = ctx.unstack(ctx.recon_download(c), c > 0)
        assert np.array_equal(got[0], pout0[c]) and np.array_equal(got[1], want1[c]), c
    ctx.close()


def test_deblock_full_size_against_the_restatement():
    hp = _hp()
    base = [c for c in deblock_cases() if (c[1], c[2]) == (256, 192)][0]
    W, H, params, parts, pin = tile_deblock_case(base, 30, 28)              # 7680 x 5376
    want = oracle_deblock(W, H, params, parts, pin)
    got, dt = _run(hp, W, H, params, parts, pin)
    for c in range(3):
        assert np.array_equal(got[c], want[c]), (c, np.argwhere(got[c] != want[c])[:5])
    print("deblock %dx%d: %.1f ms with upload of the partition data" % (W, H, dt * 1e3))


def test_deblock_after_encode_frame():
    hp = _hp()
    W, H = 200, 136
    Y, Cb, Cr = lenslet(W, H, 16, 5)
    ctx = hp.Context(W, H, slots=16)
    ctx.upload_orig(Y, Cb, Cr)
    cost, bits, dist, parts, nc = ctx.encode_frame(32, 16, 0, None, wpp=1, wavefront_lag=5)
    rec = [ctx.recon_download(c) for c in range(3)]
    want = oracle_deblock(W, H, (32, 0, 0, 0, 0), parts, rec)
    ctx.deblock_frame(parts, 32)
    got = [ctx.recon_download(c) for c in range(3)]
    ctx.close()
    assert sum(int(np.count_nonzero(a != b)) for a, b in zip(rec, want)) > 500
    for c in range(3):
        assert np.array_equal(got[c], want[c]), c
