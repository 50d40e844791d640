"""CPU, build container only: encoder-in-the-loop check of the drop-in boundary (rows a1-a6, a9-a11 and the argument mapping of
INTEGRATION.md).  oracle/_ref/TAppEncoderShim is the reference encoder with the members the C ABI replaces -- xPatternSearch,
xPatternSearchFracDIF, xPatternSearchGT, xPredInterLumaBlk / ChromaBlk, xT, xIT, xDeQuant, xRateDistOptQuant, TEncSbac::estBit, fillReferenceSamples (luma), predIntraLumaAng, calcHAD, getDistPart, xTransformSkip / xITransformSkip, xCopyYuv2SSRef, xEstimateResidualQT (the whole residual quadtree search) -- taken from
oracle/enc_shim.cpp, which forwards them to the CPU restatement; everything else is the reference's own object code.  It must
write the bitstream and the reconstruction the unmodified reference wrote (tests/golden/encoder_hop_qp32.json): then for every
call a real encode makes (real predictors and AMVP lists, sentinel regions, picture borders, AMP shapes, transform skip) the
restatement returned what the reference's member returns.  Needs /root/reference (skipped on the GPU box)."""
import hashlib
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

from hoputil import ROOT, lenslet

REF = "/root/reference"
SHIM = os.path.join(ROOT, "oracle", "_ref", "TAppEncoderShim")
CFG = os.path.join(REF, "cfg", "3DHencoder_intra_main.cfg")


def _shim():
    if not os.path.isdir(REF):
        pytest.skip("the reference tree is not present (GPU box)")
    if not os.path.exists(SHIM):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    return SHIM


def encode(exe, W, H, seed, td, extra_env=None):
    Y, Cb, Cr = lenslet(W, H, 16, seed)
    with open(os.path.join(td, "in.yuv"), "wb") as f:
        f.write(Y.astype(np.uint8).tobytes() + Cb.astype(np.uint8).tobytes() + Cr.astype(np.uint8).tobytes())
    env = dict(os.environ, HOP_SHIM_REPORT="1", **(extra_env or {}))
    r = subprocess.run([exe, "-c", CFG, "-i", "in.yuv", "-wdt", str(W), "-hgt", str(H), "-fr", "30", "-f", "1", "-q", "32", "--MIsize=16",
                        "--SEIDecodedPictureHash=1", "-b", "s.bin", "-o", "rec.yuv"], cwd=td, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    md5 = lambda n: hashlib.md5(open(os.path.join(td, n), "rb").read()).hexdigest()
    return md5("in.yuv"), md5("s.bin"), md5("rec.yuv"), r.stderr


@pytest.mark.parametrize("W,H,seed", [(64, 64, 1234), (128, 128, 1234)])
def test_shim_encoder_writes_the_reference_bitstream(W, H, seed):
    exe = _shim()
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_hop_qp32.json")))["%dx%d_seed%d" % (W, H, seed)]
    with tempfile.TemporaryDirectory() as td:
        inp, bit, rec, err = encode(exe, W, H, seed, td)
    assert inp == gold["input_md5"]
    calls = {}
    for ln in err.splitlines():
        if ln.startswith("hop shim calls:"):
            t = ln.split(":")[1].split()
            calls.update({t[i]: int(t[i + 1]) for i in range(0, len(t), 2)})
    # the replaced members really ran (a silent fall-through to the reference's definitions would also give the same bytes)
    for k in ("ss", "frac", "gt", "predY", "predC", "xT", "xIT", "dequant", "rdoq", "estBit", "fillRefs", "intraPred", "calcHAD", "distPart", "tskip", "commit", "rqt"):
        assert calls.get(k, 0) > 50, (k, calls)
    assert bit == gold["bin_md5"] and rec == gold["rec_md5"], calls
