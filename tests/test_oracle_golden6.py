"""CPU: the restatement of the deblocking filter (oracle/hop_oracle_lf.c) against the reference's own TComLoopFilter::loopFilterPic.

tests/golden/deblock_ref.npz holds what the reference's filter made of eight random pictures inside the reference encoder (oracle/make_golden22.py: random coding
quadtrees with every partition shape, transform trees, cbf, vectors around the threshold, missing reference indices; blocky planes; QP 17 - 51, slice beta / tc offsets,
chroma QP offsets; one size with partial CTUs; two 10-bit pictures, all intra, carried by an I slice of the plain configuration): partition data and planes in, planes out.  The restatement must reproduce every sample."""
import numpy as np
import pytest

from hoputil import deblock_cases, oracle_deblock


@pytest.mark.parametrize("case", deblock_cases(), ids=lambda c: "%s_%dx%d_qp%d_%dbit" % (c[0], c[1], c[2], c[3][0], c[7]))
def test_deblock_restatement_equals_the_reference_filter(case):
    key, W, H, params, parts, pin, pout, bd = case
    got = oracle_deblock(W, H, params, parts, pin, bit_depth=bd)
    changed = sum(int(np.count_nonzero(a != b)) for a, b in zip(pin, pout))
    assert changed > 1000                                                  # the fixture exercises the filter
    for c in range(3):
        assert np.array_equal(got[c], pout[c]), (key, c, np.argwhere(got[c] != pout[c])[:5])
    # slice_deblocking_filter_disabled_flag: the picture is left alone
    off = oracle_deblock(W, H, params, parts, pin, bit_depth=bd, disable=1)
    assert all(np.array_equal(a, b) for a, b in zip(off, pin))
