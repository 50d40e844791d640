#!/usr/bin/env python3
"""oracle/make_golden22.py -- tests/golden/deblock_ref.npz: inputs and outputs of the REFERENCE's own deblocking filter (TComLoopFilter::loopFilterPic inside
oracle/_ref/TAppEncoderPicCpu, built from /root/reference by oracle/Makefile.ref) on random pictures: oracle/enc_shim_pic.cpp's HOP_PIC_LF_FUZZ mode replaces a coded
picture's partition data and reconstruction by random ones (random coding quadtrees with every partition shape, transform trees, cbf, vectors around the threshold; blocky
planes), lets the reference filter them and dumps input and output (HOP_PIC_LF_DUMP).  Two picture sizes (one with partial CTUs), QPs 17 - 51, slice beta / tc offsets, chroma
QP offsets.  The run also compares the restatement (oracle/hop_oracle_lf.c) with the reference on every case.  Replayed by tests/test_oracle_golden3.py (restatement) and
tests/test_gpu_deblock.py (hop_deblock_frame).  Run in the build container."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import hop_encoder_args, lenslet, plain_encoder_args  # noqa: E402

CASES = [(200, 104, "1:32:0:0:0:0"), (200, 104, "2:22:0:0:0:0"), (200, 104, "3:37:1:-1:2:-3"), (200, 104, "4:45:-2:2:0:0"), (200, 104, "5:51:3:3:-4:5"), (200, 104, "7:17:0:0:0:0"),
         (256, 192, "8:27:0:0:0:0"), (256, 192, "9:40:0:1:1:1"),
         # 10 bit (the plain intra configuration of cfg/encoder_intra_main10.cfg carries the picture): tc and beta scaled by 4, clipping to 1023
         (136, 72, "11:27:0:0:0:0", 10), (136, 72, "12:37:1:-1:2:-3", 10)]
PART = 44
out = {}
for i, case in enumerate(CASES):
    W, H, spec = case[:3]; bd = case[3] if len(case) > 3 else 8
    with tempfile.TemporaryDirectory() as td:
        Y, Cb, Cr = lenslet(W, H, 16, 11, bitdepth=bd); dt = np.uint8 if bd == 8 else np.dtype("<u2")
        open(os.path.join(td, "in.yuv"), "wb").write(Y.astype(dt).tobytes() + Cb.astype(dt).tobytes() + Cr.astype(dt).tobytes())
        for attempt in range(6):            # the reference's GT search reads past its buffer; now and then that kills the check run (see enc_shim_pic.cpp)
            r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "TAppEncoderPicCpu")] + (hop_encoder_args(W, H) if bd == 8 else plain_encoder_args(W, H, 27, bd)), cwd=td, capture_output=True, text=True,
                               env=dict(os.environ, HOP_PIC_CHECK="1", HOP_PIC_DEBLOCK="1", HOP_PIC_LF_FUZZ=spec, HOP_PIC_LF_DUMP=os.path.join(td, "d.bin"),
                                        HOP_PIC_SPINE=os.path.join(ROOT, "oracle", "libhop_spine_cpu.so")))
            if r.returncode != 77: break
            if os.path.exists(os.path.join(td, "d.bin")): os.remove(os.path.join(td, "d.bin"))
        assert r.returncode == 0 and "deblocked picture: 0 differences" in r.stderr, r.stderr[-800:]
        raw = open(os.path.join(td, "d.bin"), "rb").read()
    hd = np.frombuffer(raw, "<i4", 10); assert hd[0] == W and hd[1] == H and hd[8] == PART and hd[9] == bd, hd
    n = int(hd[7]); o = 40
    parts = np.frombuffer(raw, np.uint8, n * 256 * PART, o).reshape(n, 256, PART); o += n * 256 * PART
    planes = []
    for k in range(6):
        cnt = W * H if k % 3 == 0 else W * H // 4
        planes.append(np.frombuffer(raw, "<i2", cnt, o).astype(np.uint8 if bd == 8 else np.uint16)); o += cnt * 2
    assert o == len(raw)
    key = "c%d" % i
    out[key + "/geo"] = hd[:7].astype(np.int32); out[key + "/bd"] = np.int32(bd); out[key + "/parts"] = parts
    for k, nm in enumerate(("y", "cb", "cr")): out[key + "/in_" + nm] = planes[k]; out[key + "/out_" + nm] = planes[3 + k]
    print(key, W, H, spec, bd, [ln for ln in r.stderr.splitlines() if "fuzz" in ln][0])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "deblock_ref.npz"), **out)
print(os.path.getsize(os.path.join(ROOT, "tests", "golden", "deblock_ref.npz")), "bytes")
