#!/usr/bin/env python3
"""oracle/make_golden18.py -- TEST INFRASTRUCTURE.  Samples the residual-free candidates (encodeResAndCalcRdInterCU with bSkipRes, TEncSearch.cpp:6635-6668) of two real
encodes (64x64 golden lenslet and the 64x64 sharp-edged frame) into tests/golden/encoder_cuskip_calls.npz: the shim encoder (oracle/enc_shim.cpp; its bitstream equals
the unmodified reference's, tests/test_encoder_shim.py) runs with HOP_SHIM_TRACE_CUSKIP; calls are kept spread over CU size, skip context and merge index: parameters,
prediction and original planes of the CU, coder and CU-level contexts in and out, bits, the three distortions, the cost.  Replayed by tests/test_oracle_golden5.py
(restatement) and tests/test_gpu_tq_intra.py (hop_inter_cu_skip on the GPU).  Needs /root/reference (build container)."""
import os, struct, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import lenslet, sharp_frame
from make_golden9 import CFG, CODER
PER_KIND = 2


def calls(frame):
    exe = os.path.join(ROOT, "oracle", "_ref", "TAppEncoderShim")
    Y, Cb, Cr = frame
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "in.yuv"), "wb").write(Y.astype(np.uint8).tobytes() + Cb.astype(np.uint8).tobytes() + Cr.astype(np.uint8).tobytes())
        tr = os.path.join(td, "t.bin")
        r = subprocess.run([exe, "-c", "/root/reference/cfg/3DHencoder_intra_main.cfg", "-i", "in.yuv", "-wdt", "64", "-hgt", "64", "-fr", "30", "-f", "1", "-q", "32",
                            "--MIsize=16", "-b", "s.bin", "-o", "rec.yuv"], cwd=td, capture_output=True, text=True, env=dict(os.environ, HOP_SHIM_TRACE_CUSKIP=tr))
        assert r.returncode == 0, r.stderr[-2000:]
        b = open(tr, "rb").read()
    recs, o = [], 0
    while o < len(b):
        cfg = np.frombuffer(b, CFG, 1, o)[0]; o += CFG.itemsize
        nd = struct.unpack_from("<4i", b, o); o += 16
        cu = 1 << int(cfg["log2_cu"]); n = cu * cu * 3 // 2
        pred = np.frombuffer(b, "<i2", n, o).copy(); o += 2 * n
        org = np.frombuffer(b, "<i2", n, o).copy(); o += 2 * n
        cin = np.frombuffer(b, CODER, 1, o)[0]; o += 160
        cuin = np.frombuffer(b, "u1", 16, o).copy(); o += 16
        cout = np.frombuffer(b, CODER, 1, o)[0]; o += 160
        cuout = np.frombuffer(b, "u1", 16, o).copy(); o += 16
        o4 = struct.unpack_from("<4I", b, o); o += 16
        cost = struct.unpack_from("<d", b, o)[0]; o += 8
        recs.append(dict(cfg=cfg, nd=nd, pred=pred, org=org, cin=cin, cuin=cuin, cout=cout, cuout=cuout, o4=o4, cost=cost))
    assert o == len(b)
    print(len(recs), "calls")
    return recs


def main():
    recs = calls(lenslet(64, 64, 16, 1234)) + calls(sharp_frame(64, 64, 77))
    rng = np.random.default_rng(18)
    groups = {}
    for r in recs:
        groups.setdefault((int(r["cfg"]["log2_cu"]), r["nd"][0], r["nd"][1]), []).append(r)
    keep = []
    for k in sorted(groups):
        L = groups[k]
        keep += [L[i] for i in rng.permutation(len(L))[:PER_KIND]]
    print(len(groups), "kinds ->", len(keep), "calls; sizes", sorted(set(int(r["cfg"]["log2_cu"]) for r in keep)))
    path = os.path.join(ROOT, "tests", "golden", "encoder_cuskip_calls.npz")
    np.savez_compressed(path, cfg=np.array([r["cfg"] for r in keep]), nd=np.array([r["nd"] for r in keep], np.int32), pred=np.concatenate([r["pred"] for r in keep]),
                        org=np.concatenate([r["org"] for r in keep]), cin=np.array([r["cin"] for r in keep]), cuin=np.stack([r["cuin"] for r in keep]),
                        cout=np.array([r["cout"] for r in keep]), cuout=np.stack([r["cuout"] for r in keep]), o4=np.array([r["o4"] for r in keep], np.uint32),
                        cost=np.array([r["cost"] for r in keep]))
    print(len(keep), "calls ->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
