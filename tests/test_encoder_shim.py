"""CPU, build container only: encoder-in-the-loop check of the drop-in boundary (rows a1-a6, a9-a11 and the argument mapping of
INTEGRATION.md).  oracle/_ref/TAppEncoderShim is the reference encoder with the members the C ABI replaces -- xPatternSearch,
xPatternSearchFracDIF, xPatternSearchGT, xPredInterLumaBlk / ChromaBlk, xT, xIT, xDeQuant, xRateDistOptQuant, TEncSbac::estBit, fillReferenceSamples (luma), predIntraLumaAng, calcHAD, getDistPart, xTransformSkip / xITransformSkip, xCopyYuv2SSRef, xEstimateResidualQT (the whole residual quadtree search), xAddSymbolBitsInter (the CU-level syntax bits), xModeBitsIntra, xUpdateCandList, xGetIntraBitsQT, predIntraChromaAng (and fillReferenceSamples for the chroma planes too) -- taken from
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

from hoputil import ROOT, lenslet, sharp_frame

REF = "/root/reference"
SHIM = os.path.join(ROOT, "oracle", "_ref", "TAppEncoderShim")
CFG = os.path.join(REF, "cfg", "3DHencoder_intra_main.cfg")


def _shim():
    if not os.path.isdir(REF):
        pytest.skip("the reference tree is not present (GPU box)")
    if not os.path.exists(SHIM):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    return SHIM


def encode(exe, W, H, seed, td, extra_env=None, sharp=False):
    Y, Cb, Cr = sharp_frame(W, H, seed) if sharp else lenslet(W, H, 16, seed)
    with open(os.path.join(td, "in.yuv"), "wb") as f:
        f.write(Y.astype(np.uint8).tobytes() + Cb.astype(np.uint8).tobytes() + Cr.astype(np.uint8).tobytes())
    env = dict(os.environ, HOP_SHIM_REPORT="1", **(extra_env or {}))
    for attempt in range(6):          # the reference's own GT search reads past its reference picture buffer; now and then that kills the UNMODIFIED encoder (SIGSEGV): repeat such a run
        r = subprocess.run([exe, "-c", CFG, "-i", "in.yuv", "-wdt", str(W), "-hgt", str(H), "-fr", "30", "-f", "1", "-q", "32", "--MIsize=16",
                            "--SEIDecodedPictureHash=1", "-b", "s.bin", "-o", "rec.yuv"], cwd=td, capture_output=True, text=True, env=env)
        if not (r.returncode == -11 and os.path.basename(exe) == "TAppEncoderRef"):
            break
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    md5 = lambda n: hashlib.md5(open(os.path.join(td, n), "rb").read()).hexdigest()
    return md5("in.yuv"), md5("s.bin"), md5("rec.yuv"), r.stderr


LOW = ("ss", "frac", "gt", "predY", "predC", "xT", "xIT", "dequant", "rdoq", "estBit", "fillRefs", "distPart", "commit", "rqt", "cuBits", "intraBits")
# the composite restatements stand in for members that call other replaced members; HOP_SHIM_ORIG hands the named composites back to the reference's own
# definitions, so every level of the stack is reached (and counted) by some run
LEVELS = {"": ("ss", "frac", "gt", "predY", "predC", "distPart", "commit", "rqt", "cuBits", "intraSearch", "chromaSearch", "intraCu", "cuSkip"),   # transforms, RDOQ, estBit, prediction: all inside the composites here
          "estIntraPredQT,estIntraPredChromaQT": LOW + ("intraRqt", "modeBits", "candList", "intraPred", "calcHAD", "chromaPred"),
          "estIntraPredQT,estIntraPredChromaQT,xRecurIntraCodingQT": LOW + ("modeBits", "candList", "intraPred", "calcHAD", "chromaPred", "tskip")}


@pytest.mark.parametrize("orig", list(LEVELS))
@pytest.mark.parametrize("W,H,seed", [(64, 64, 1234), (128, 128, 1234)])
def test_shim_encoder_writes_the_reference_bitstream(W, H, seed, orig):
    exe = _shim()
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_hop_qp32.json")))["%dx%d_seed%d" % (W, H, seed)]
    with tempfile.TemporaryDirectory() as td:
        inp, bit, rec, err = encode(exe, W, H, seed, td, extra_env={"HOP_SHIM_ORIG": orig})
    assert inp == gold["input_md5"]
    calls = {}
    for ln in err.splitlines():
        if ln.startswith("hop shim calls:"):
            t = ln.split(":")[1].split()
            calls.update({t[i]: int(t[i + 1]) for i in range(0, len(t), 2)})
    # the replaced members really ran (a silent fall-through to the reference's definitions would also give the same bytes)
    for k in LEVELS[orig]:
        assert calls.get(k, 0) > 50, (k, calls)
    for k in ("intraSearch", "chromaSearch", "intraRqt"):
        if k not in LEVELS[orig]: assert calls.get(k, 0) == 0, (k, calls)
    assert bit == gold["bin_md5"] and rec == gold["rec_md5"], calls


def test_shim_encoder_on_sharp_content():
    """a frame on which the 4x4 transform-skip variant wins in nearly every residual quadtree (the lenslets never choose it): the shim
    encoder against the unmodified reference encoder, both run here"""
    exe = _shim()
    ref = os.path.join(ROOT, "oracle", "_ref", "TAppEncoderRef")
    with tempfile.TemporaryDirectory() as ta, tempfile.TemporaryDirectory() as tb:
        a = encode(ref, 64, 64, 77, ta, sharp=True)
        b = encode(exe, 64, 64, 77, tb, sharp=True, extra_env={"HOP_SHIM_TRACE_RQT": os.path.join(tb, "rqt.bin")})
        raw = open(os.path.join(tb, "rqt.bin"), "rb").read()
    assert a[:3] == b[:3]
    # the trace of the residual-quadtree calls (layout: oracle/make_golden9.py) must show the transform-skip arrays in use
    o = n = ts = 0
    while o < len(raw):
        cu = 1 << int(np.frombuffer(raw, "<i4", 1, o)[0])
        o += 104 + 160 + 160 + cu * cu * 3 + 8 + 16
        ts += int(np.frombuffer(raw, "u1", 768, o + 1024).any()); n += 1
        o += 256 * 7 + cu * cu * 6
    assert n > 300 and ts > n // 2, (n, ts)


def _encode_plain(exe, cfg, W, H, seed, qp, bd, td):
    Y, Cb, Cr = lenslet(W, H, 16, seed, bitdepth=bd)
    dt = np.uint8 if bd == 8 else np.dtype("<u2")
    with open(os.path.join(td, "in.yuv"), "wb") as f:
        f.write(Y.astype(dt).tobytes() + Cb.astype(dt).tobytes() + Cr.astype(dt).tobytes())
    r = subprocess.run([exe, "-c", os.path.join(REF, "cfg", cfg), "-i", "in.yuv", "-wdt", str(W), "-hgt", str(H), "-fr", "30", "-f", "1", "-q", str(qp), "--InputBitDepth=%d" % bd,
                        "--SEIDecodedPictureHash=1", "-b", "s.bin", "-o", "rec.yuv"], cwd=td, capture_output=True, text=True, env=dict(os.environ, HOP_SHIM_REPORT="1"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    md5 = lambda n: hashlib.md5(open(os.path.join(td, n), "rb").read()).hexdigest()
    return md5("s.bin"), md5("rec.yuv"), r.stderr


@pytest.mark.parametrize("cfg,bd,qp", [("encoder_intra_main.cfg", 8, 32), ("encoder_intra_main10.cfg", 10, 22), ("encoder_intra_main10.cfg", 10, 27),
                                       ("encoder_intra_main10.cfg", 10, 32), ("encoder_intra_main10.cfg", 10, 37)])
def test_shim_encoder_plain_intra_configurations(cfg, bd, qp):
    """BASELINE configs 1 and 4: the plain HM intra configurations (I slices; 8 bit, and 10 bit at the four QPs of the RD sweep): the shim encoder -- composite restatements of
    the intra searches, the intra CU's bit count, RDOQ, transforms, CABAC estimation in place of the reference's members -- against the unmodified encoder run beside it."""
    exe = _shim()
    ref = os.path.join(ROOT, "oracle", "_ref", "TAppEncoderRef")
    with tempfile.TemporaryDirectory() as ta, tempfile.TemporaryDirectory() as tb:
        a = _encode_plain(ref, cfg, 128, 128, 9, qp, bd, ta)
        b = _encode_plain(exe, cfg, 128, 128, 9, qp, bd, tb)
    calls = {}
    for ln in b[2].splitlines():
        if ln.startswith("hop shim calls:"):
            t = ln.split(":")[1].split()
            calls.update({t[i]: int(t[i + 1]) for i in range(0, len(t), 2)})
    for k in ("intraSearch", "chromaSearch", "intraCu"):
        assert calls.get(k, 0) > 100, (k, calls)
    assert a[:2] == b[:2], calls
