"""CPU, build container only: the picture-level reference-side binding (oracle/enc_shim_pic.cpp) over the CPU spine.

oracle/_ref/TAppEncoderPicCpu is the reference encoder with TEncCu::compressCU alone replaced: a picture goes through the spine once, and every compressCU call only fills
the CTU's TComDataCU from what the spine exports (hop_cu_part, the levels, the reconstruction, cost / bits / distortion).  The counting pass of compressSlice, encodeSlice,
deblocking, SAO, the picture hash and the bitstream writer are the reference's own object code.  The bitstream and the reconstruction must be, byte for byte, those of the
unmodified reference encoder (tests/golden/encoder_hop_pic.json, made by oracle/make_golden21.py): then what the spine exports is complete and right down to the last bit
the entropy coder writes -- a wrong level, cbf, mvd, merge index or transform-skip flag anywhere in the picture changes the md5.  tests/test_gpu_encoder_pic.py runs the same
binding over libhophip.so on the GPU box.  Needs /root/reference to build (skipped on the GPU box)."""
import hashlib
import json
import os
import re
import subprocess
import tempfile

import pytest

from hoputil import PIC_CASES, ROOT, hop_encoder_args, pic_case_args, pic_case_input

REF = "/root/reference"
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_hop_pic.json")))
GOLD.update(json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_hop_pic_mi15.json"))))     # the pictures coded with --MIsize=15 (oracle/make_golden24.py)


def run_binding(exe, key, env, decoder=None):
    c = PIC_CASES[key]
    raw = pic_case_input(c)
    assert hashlib.md5(raw).hexdigest() == GOLD[key]["input_md5"]
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "in.yuv"), "wb").write(raw)
        r = subprocess.run([exe] + pic_case_args(c), cwd=td, capture_output=True, text=True, env=dict(os.environ, HOP_SHIM_REPORT="1", **env))
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        md5 = lambda n: hashlib.md5(open(os.path.join(td, n), "rb").read()).hexdigest()
        got = {"bin_md5": md5("s.bin"), "rec_md5": md5("rec.yuv"), "bin_bytes": os.path.getsize(os.path.join(td, "s.bin"))}
        dec_report = {}
        if decoder:             # the stream just written, through a decoder: the decoded pictures are the encoder's reconstruction
            d = subprocess.run([decoder, "-b", "s.bin", "-o", "dec.yuv"], cwd=td, capture_output=True, text=True, env=dict(os.environ, HOP_SHIM_REPORT="1"))
            assert d.returncode == 0, d.stdout[-1500:] + d.stderr[-1500:]
            assert md5("dec.yuv") == got["rec_md5"], "decoded pictures differ from the encoder's reconstruction"
            assert "ERROR" not in d.stdout and "***ERROR***" not in d.stderr
            for ln in d.stderr.splitlines():
                if ln.startswith("hop dec binding:"):
                    t = ln.split(":")[1].split(); dec_report = {"dec_" + t[i]: int(t[i + 1]) for i in range(0, len(t), 2)}
    # the PSNR the binding computed for each final picture (HOP_PIC_SAO) against the three numbers the reference prints per picture
    ours = [ln.split(":")[1].split() for ln in r.stderr.splitlines() if ln.startswith("hop pic psnr:")]
    theirs = [re.findall(r"\[Y\s+([0-9.]+) dB\s+U\s+([0-9.]+) dB\s+V\s+([0-9.]+) dB\]", ln)[0] for ln in r.stdout.splitlines() if ln.startswith("POC")]
    if ours:
        assert [tuple(o) for o in ours] == [tuple(t) for t in theirs], (ours, theirs)
    rep = [ln for ln in r.stderr.splitlines() if ln.startswith("hop pic binding:")]
    assert len(rep) == 1, r.stderr[-1500:]
    t = rep[0].split(":")[1].split()
    return got, dict({t[i]: int(t[i + 1]) for i in range(0, len(t), 2)}, **dec_report)


def check(key, got, counts):
    c = PIC_CASES[key]
    # the replaced member really ran, once per CTU of every picture (a fall-through to the reference's own compressCU would write the same bytes)
    assert counts["pictures"] == c["frames"] and counts["ctus"] == c["frames"] * ((c["W"] + 63) // 64) * ((c["H"] + 63) // 64) and counts["candidates"] > (20 if "plain" in c else 100) * counts["ctus"], counts
    want = {k: GOLD[key][k] for k in got}
    assert got == want, key


@pytest.mark.parametrize("key", ["64x64_raster", "192x128_wpp", "200x104_raster", "128x64_2frames", "136x72_plain8", "200x136_seed5_mi15", "192x128_seed7_mi15_wpp"])
def test_reference_encoder_over_the_cpu_spine_writes_the_reference_bitstream(key):
    if not os.path.isdir(REF):
        pytest.skip("the reference tree is not present (GPU box)")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libhop_spine_cpu.so"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref", "-j4", "_ref/TAppEncoderPicCpu"], stdout=subprocess.DEVNULL)
    # (and round the loop: the reference DECODER reads the stream back to the reconstruction)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref", "-j4", "_ref/TAppDecoderRef"], stdout=subprocess.DEVNULL)
    got, counts = run_binding(os.path.join(ROOT, "oracle", "_ref", "TAppEncoderPicCpu"), key, {"HOP_PIC_SPINE": os.path.join(ROOT, "oracle", "libhop_spine_cpu.so")},
                              decoder=os.path.join(ROOT, "oracle", "_ref", "TAppDecoderRef") if key in ("200x104_raster", "192x128_wpp") else None)   # (HOP streams: the fork's decoder does not survive a plain I-slice stream, not even its own encoder's)
    check(key, got, counts)


def test_reference_encoder_with_the_restated_loop_filters():
    """HOP_PIC_DEBLOCK: loopFilterPic replaced as well (here by the restatement, oracle/hop_oracle_lf.c): same bitstream, same reconstruction"""
    if not os.path.isdir(REF):
        pytest.skip("the reference tree is not present (GPU box)")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libhop_spine_cpu.so"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref", "-j4", "_ref/TAppEncoderPicCpu"], stdout=subprocess.DEVNULL)
    got, counts = run_binding(os.path.join(ROOT, "oracle", "_ref", "TAppEncoderPicCpu"), "200x104_raster", {"HOP_PIC_SPINE": os.path.join(ROOT, "oracle", "libhop_spine_cpu.so"), "HOP_PIC_DEBLOCK": "1"})
    assert counts["deblocked"] == 1
    check("200x104_raster", got, counts)
    # ... and SAOProcess (restated statistics / offsetting around the product's decision, hop_sao_decide)
    got, counts = run_binding(os.path.join(ROOT, "oracle", "_ref", "TAppEncoderPicCpu"), "128x64_2frames", {"HOP_PIC_SPINE": os.path.join(ROOT, "oracle", "libhop_spine_cpu.so"), "HOP_PIC_DEBLOCK": "1", "HOP_PIC_SAO": "1"})
    assert counts["deblocked"] == 2 and counts["sao"] == 2
    check("128x64_2frames", got, counts)
    # ... and at 10 bit (cfg/encoder_intra_main10.cfg, I slice): the tc / beta scaling of the deblocking filter, SAO's 31-step offsets and its distortion shift
    got, counts = run_binding(os.path.join(ROOT, "oracle", "_ref", "TAppEncoderPicCpu"), "136x72_plain10", {"HOP_PIC_SPINE": os.path.join(ROOT, "oracle", "libhop_spine_cpu.so"), "HOP_PIC_DEBLOCK": "1", "HOP_PIC_SAO": "1"})
    assert counts["deblocked"] == 1 and counts["sao"] == 1
    check("136x72_plain10", got, counts)


def test_golden_of_the_binding_agrees_with_the_configuration_file():
    """the options of hoputil.HOP_ENCODER_OPTIONS select the configuration of cfg/3DHencoder_intra_main.cfg: the two goldens made with one and with the other agree where they
    hold the same picture"""
    old = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_hop_qp32.json")))
    for a, b in (("64x64_raster", "64x64_seed1234"), ("192x128_raster", "192x128_seed7")):
        assert all(GOLD[a][k] == old[b][k] for k in ("input_md5", "bin_md5", "rec_md5")), a


@pytest.mark.parametrize("key", ["200x104_raster", "192x128_wpp", "136x72_plain10"])
def test_binding_fields_against_the_reference_ctu_by_ctu(key):
    """HOP_PIC_CHECK: the reference's own compressCU codes every CTU before the binding fills it, and each field the binding writes -- every per-partition array, every
    level, cost / bits / distortion, and the carried fraction of the RD coder (hop_rd_fraction_download) -- is compared with what the reference left"""
    if not os.path.isdir(REF):
        pytest.skip("the reference tree is not present (GPU box)")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libhop_spine_cpu.so"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref", "-j4", "_ref/TAppEncoderPicCpu"], stdout=subprocess.DEVNULL)
    c = PIC_CASES[key]
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "in.yuv"), "wb").write(pic_case_input(c))
        for attempt in range(6):
            r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "TAppEncoderPicCpu")] + pic_case_args(c), cwd=td, capture_output=True, text=True,
                               env=dict(os.environ, HOP_PIC_CHECK="1", HOP_PIC_SPINE=os.path.join(ROOT, "oracle", "libhop_spine_cpu.so")))
            # the reference's GT search reads past its reference picture buffer and dies when that page is unmapped (in its own xPatternSearchGT; the unmodified encoder does
            # the same now and then): such a run says nothing about the binding and is repeated
            if not (r.returncode == 77 and "the reference's own code faulted" in r.stderr and "xPatternSearchGT" in r.stderr):
                break
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [ln for ln in r.stderr.splitlines() if ln.startswith("hop pic check:")]
    n = ((c["W"] + 63) // 64) * ((c["H"] + 63) // 64)
    assert lines == ["hop pic check: CTU %d: 0 differences" % a for a in range(n)], lines[:20]
