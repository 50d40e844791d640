#!/usr/bin/env python3
"""oracle/make_golden21.py -- tests/golden/encoder_hop_pic.json: md5 of the bitstream and of the reconstruction the UNMODIFIED reference encoder (oracle/_ref/TAppEncoderRef,
built from /root/reference by oracle/Makefile.ref) writes for the pictures the picture-level binding is tested on (tests/test_encoder_pic.py here, tests/test_gpu_encoder_pic.py
on the GPU box): the HOP configuration given as command-line options (tests/hoputil.py: HOP_ENCODER_OPTIONS), raster order and WaveFrontSynchro, picture sizes that are
and are not multiples of the CTU, one and two frames.  Run in the build container."""
import hashlib, json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import PIC_CASES, pic_case_args, pic_case_input  # noqa: E402

out = {}
for key, c in PIC_CASES.items():
    if "pitch" in c: continue                      # the --MIsize=15 pictures have their own golden (oracle/make_golden24.py)
    with tempfile.TemporaryDirectory() as td:
        raw = pic_case_input(c)
        open(os.path.join(td, "in.yuv"), "wb").write(raw)
        for attempt in range(6):                 # the reference's GT search reads past its reference picture buffer; now and then that kills the process (SIGSEGV in xPatternSearchGT)
            r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "TAppEncoderRef")] + pic_case_args(c), cwd=td, capture_output=True, text=True)
            if r.returncode != -11: break
        assert r.returncode == 0, r.stdout[-2000:]
        md5 = lambda n: hashlib.md5(open(os.path.join(td, n), "rb").read()).hexdigest()
        out[key] = {"input_md5": hashlib.md5(raw).hexdigest(), "bin_md5": md5("s.bin"), "rec_md5": md5("rec.yuv"), "bin_bytes": os.path.getsize(os.path.join(td, "s.bin"))}
        print(key, out[key])
        if "plain" in c:      # the options must select what the configuration file selects: the same run with -c <the reference's file>
            cfg = "/root/reference/cfg/" + ("encoder_intra_main.cfg" if c["plain"][0] == 8 else "encoder_intra_main10.cfg")
            r2 = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "TAppEncoderRef"), "-c", cfg, "-i", "in.yuv", "-wdt", str(c["W"]), "-hgt", str(c["H"]), "-fr", "30", "-f", "1", "-q", str(c["plain"][1]),
                                 "--InputBitDepth=%d" % c["plain"][0], "--SEIDecodedPictureHash=1", "-b", "s2.bin", "-o", "rec2.yuv"], cwd=td, capture_output=True, text=True)
            assert r2.returncode == 0 and md5("s2.bin") == out[key]["bin_md5"] and md5("rec2.yuv") == out[key]["rec_md5"], (key, "options and configuration file disagree", r2.stdout[-500:])
            print(key, "equals the run with", cfg)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "encoder_hop_pic.json"), "w"), indent=1, sort_keys=True)
