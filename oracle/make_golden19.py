#!/usr/bin/env python3
"""oracle/make_golden19.py -- TEST INFRASTRUCTURE.  Golden vectors of the RD spine from the reference encoder: for five synthetic frames (lenslets 64x64,
128x128, 192x128, 200x136 with picture-boundary CTUs, the sharp-edged 64x64 frame, and 192x128 / 448x192 with WaveFrontSynchro; plus 136x72 frames in the plain intra configurations -- cfg/encoder_intra_main.cfg 8 bit QP 32 and
encoder_intra_main10.cfg 10 bit QP 22 / 27 / 32 / 37, I slices --) the shim encoder (oracle/enc_shim.cpp -- its bitstream equals the unmodified
reference's, tests/test_encoder_shim.py) runs with its observers on: HOP_SHIM_TRACE_BEST = every candidate that reaches TEncCu::xCheckBestMode (depth, position, mode,
partition, skip / merge flags, bits, distortion, cost), HOP_SHIM_TRACE_CTU = every CTU's finished TComDataCU (cost, bits, distortion, per-partition depth / mode / partition /
flags / directions / transform depth / cbf / vectors).  The per-CTU costs are cross-checked against cost.csv of the UNMODIFIED encoder (tests/golden/encoder_hop_qp32.json)
where that holds the frame.  Written to tests/golden/encoder_spine.npz; replayed by tests/test_spine_cpu.py (the spine over the CPU restatement) and tests/test_gpu_spine.py
(hop_encode_frame on the GPU).  Needs /root/reference (build container)."""
import json, os, sys, tempfile, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from spine_check import run_reference, read_ctu_trace

FRAMES = [(64, 64, 1234, False, False), (128, 128, 1234, False, False), (192, 128, 7, False, False), (200, 136, 5, False, False), (64, 64, 77, True, False),
          # the reference run with --WaveFrontSynchro=1 --WaveFrontSubstreams=<CTU rows>: what the wavefront mode of the spine must reproduce
          (192, 128, 7, False, True), (448, 192, 3, False, True)]

PLAIN = [("encoder_intra_main.cfg", 8, 32), ("encoder_intra_main10.cfg", 10, 22), ("encoder_intra_main10.cfg", 10, 27), ("encoder_intra_main10.cfg", 10, 32), ("encoder_intra_main10.cfg", 10, 37)]

def run_plain(cfg, W, H, seed, qp, bd, td):
    """the plain HM intra configurations (I slice; BASELINE configs 1 and 4): 8-bit, and 10-bit samples from a 16-bit file (--InputBitDepth=10)"""
    import subprocess
    from hoputil import lenslet
    Y, Cb, Cr = lenslet(W, H, 16, seed, bitdepth=bd)
    dt = np.uint8 if bd == 8 else np.dtype("<u2")
    open(os.path.join(td, "in.yuv"), "wb").write(Y.astype(dt).tobytes() + Cb.astype(dt).tobytes() + Cr.astype(dt).tobytes())
    env = dict(os.environ, HOP_SHIM_TRACE_BEST=os.path.join(td, "best.txt"), HOP_SHIM_TRACE_CTU=os.path.join(td, "ctu.bin"))
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "TAppEncoderShim"), "-c", "/root/reference/cfg/" + cfg, "-i", "in.yuv", "-wdt", str(W), "-hgt", str(H), "-fr", "30", "-f", "1",
                        "-q", str(qp), "--InputBitDepth=%d" % bd, "-b", "s.bin", "-o", "rec.yuv"], cwd=td, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-1500:]


def main():
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_hop_qp32.json")))
    out = {}
    for W, H, seed, sharp, wpp in FRAMES:
        key = "%dx%d_seed%d%s%s" % (W, H, seed, "_sharp" if sharp else "", "_wpp" if wpp else "")
        with tempfile.TemporaryDirectory() as td:
            run_reference(W, H, seed, sharp, td, wpp=wpp)
            text = open(os.path.join(td, "best.txt"), "rb").read()
            ctu = read_ctu_trace(os.path.join(td, "ctu.bin"))
            csv = open(os.path.join(td, "cost.csv")).read() if os.path.exists(os.path.join(td, "cost.csv")) else None
        g = gold.get("%dx%d_seed%d" % (W, H, seed)) if not (sharp or wpp) else None
        if g:                                                    # cost.csv of the unmodified reference encoder: one line per picture: POC;cost;cost;... in coding order (TEncSlice.cpp:183-191)
            ref_costs = [float(v) for v in g["cost_csv"].strip().split(";")[1:]]
            assert ref_costs == [float(c) for c in ctu["cost"]], (key, ref_costs, list(ctu["cost"]))
        out[key + "/cost"] = ctu["cost"].astype(np.float64); out[key + "/bits"] = ctu["bits"].astype(np.uint32); out[key + "/dist"] = ctu["dist"].astype(np.uint32)
        out[key + "/parts"] = ctu["p"].astype(np.int16)
        out[key + "/trace"] = np.frombuffer(zlib.compress(text, 9), np.uint8)
        print(key, len(ctu), "CTUs", text.count(b"\n"), "candidates", "cost.csv of the unmodified encoder matched" if g else "")
    for cfg, bd, qp in PLAIN:
        W, H, seed = 136, 72, 9
        key = "plain%d_qp%d_%dx%d_seed%d" % (bd, qp, W, H, seed)
        with tempfile.TemporaryDirectory() as td:
            run_plain(cfg, W, H, seed, qp, bd, td)
            text = open(os.path.join(td, "best.txt"), "rb").read()
            ctu = read_ctu_trace(os.path.join(td, "ctu.bin"))
        out[key + "/cost"] = ctu["cost"].astype(np.float64); out[key + "/bits"] = ctu["bits"].astype(np.uint32); out[key + "/dist"] = ctu["dist"].astype(np.uint32)
        out[key + "/parts"] = ctu["p"].astype(np.int16)
        out[key + "/trace"] = np.frombuffer(zlib.compress(text, 9), np.uint8)
        print(key, len(ctu), "CTUs", text.count(b"\n"), "candidates")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "encoder_spine.npz"), **out)

if __name__ == "__main__":
    main()
