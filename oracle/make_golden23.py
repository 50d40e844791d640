#!/usr/bin/env python3
"""oracle/make_golden23.py -- tests/golden/sao_ref.npz: inputs and outputs of the REFERENCE's own SAO encoder (TEncSampleAdaptiveOffset::SAOProcess inside
oracle/_ref/TAppEncoderPicCpu) on random pictures: oracle/enc_shim_pic.cpp's HOP_PIC_SAO_FUZZ mode replaces a coded picture's original and deblocked planes by random ones
(per CTU: no error, band-dependent shifts, ringing along one of the four edge directions, noise; often the kind of the left neighbour), scales the lambdas, lets the
reference decide and offset, and dumps (HOP_PIC_SAO_DUMP) the planes in, the slice's lambdas / QP / type, the fraction the RD coder carried, and the reference's
statistics, parameters and planes out.  The run also compares the library-side path (restated statistics / offsetting, the product's decision) with the reference on every
case.  Replayed by tests/test_oracle_golden7.py (CPU) and tests/test_gpu_sao.py (hop_sao_stats / hop_sao_frame).  Run in the build container."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import hop_encoder_args, lenslet, plain_encoder_args  # noqa: E402

CASES = [(200, 104, "1:100"), (200, 104, "2:30"), (200, 104, "4:10"), (200, 104, "6:3"), (256, 192, "7:20"), (256, 192, "8:100"), (256, 192, "9:1"),
         # 10 bit (an I slice of cfg/encoder_intra_main10.cfg carries the picture): offsets up to 31, the distortion shift of 4 bits, 32 bands of 32 values
         (136, 72, "11:30", 10), (136, 72, "12:100", 10), (136, 72, "13:5", 10)]
PAR = 36
out = {}
for i, case in enumerate(CASES):
    W, H, spec = case[:3]; bd = case[3] if len(case) > 3 else 8
    with tempfile.TemporaryDirectory() as td:
        Y, Cb, Cr = lenslet(W, H, 16, 11, bitdepth=bd); dt = np.uint8 if bd == 8 else np.dtype("<u2")
        open(os.path.join(td, "in.yuv"), "wb").write(Y.astype(dt).tobytes() + Cb.astype(dt).tobytes() + Cr.astype(dt).tobytes())
        for attempt in range(6):            # the reference's GT search reads past its buffer; now and then that kills the check run (see enc_shim_pic.cpp)
            r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "TAppEncoderPicCpu")] + (hop_encoder_args(W, H) if bd == 8 else plain_encoder_args(W, H, 27, bd)), cwd=td, capture_output=True, text=True,
                               env=dict(os.environ, HOP_PIC_CHECK="1", HOP_PIC_SAO="1", HOP_PIC_SAO_FUZZ=spec, HOP_PIC_SAO_DUMP=os.path.join(td, "d.bin"),
                                        HOP_PIC_SPINE=os.path.join(ROOT, "oracle", "libhop_spine_cpu.so")))
            if r.returncode != 77: break
        assert r.returncode == 0 and "SAO: 0 differences" in r.stderr, r.stderr[-800:]
        raw = open(os.path.join(td, "d.bin"), "rb").read()
    hd = np.frombuffer(raw, "<i4", 7); assert hd[0] == W and hd[1] == H and hd[6] == bd, hd
    n = int(hd[2]); o = 28
    lam = np.frombuffer(raw, "<f8", 3, o); o += 24
    planes = []
    for k in range(6):
        cnt = W * H if k % 3 == 0 else W * H // 4
        planes.append(np.frombuffer(raw, "<i2", cnt, o).astype(np.uint8 if bd == 8 else np.uint16)); o += cnt * 2
    stats = np.frombuffer(raw, "<i4", n * 3 * 5 * 32 * 2, o).reshape(n, 3, 5, 32, 2); o += stats.size * 4
    par = np.frombuffer(raw, np.int8, n * 3 * PAR, o).reshape(n, 3, PAR); o += par.size
    outp = []
    for k in range(3):
        cnt = W * H if k == 0 else W * H // 4
        outp.append(np.frombuffer(raw, "<i2", cnt, o).astype(np.uint8 if bd == 8 else np.uint16)); o += cnt * 2
    assert o == len(raw), (o, len(raw))
    key = "c%d" % i
    out[key + "/geo"] = hd[:6].astype(np.int32); out[key + "/bd"] = np.int32(bd); out[key + "/lambda"] = lam.copy(); out[key + "/stats"] = stats; out[key + "/coded"] = par
    for k, nm in enumerate(("y", "cb", "cr")): out[key + "/org_" + nm] = planes[k]; out[key + "/in_" + nm] = planes[3 + k]; out[key + "/out_" + nm] = outp[k]
    print(key, W, H, spec, bd, [ln for ln in r.stderr.splitlines() if "fuzz" in ln][0])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sao_ref.npz"), **out)
print(os.path.getsize(os.path.join(ROOT, "tests", "golden", "sao_ref.npz")), "bytes")
