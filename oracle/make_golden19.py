#!/usr/bin/env python3
"""oracle/make_golden19.py -- TEST INFRASTRUCTURE.  Golden vectors of the RD spine from the reference encoder: for five synthetic frames (lenslets 64x64,
128x128, 192x128, 200x136 with picture-boundary CTUs, the sharp-edged 64x64 frame, and 192x128 / 448x192 with WaveFrontSynchro) the shim encoder (oracle/enc_shim.cpp -- its bitstream equals the unmodified
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
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "encoder_spine.npz"), **out)

if __name__ == "__main__":
    main()
