#!/usr/bin/env python3
"""oracle/make_golden17.py -- TEST INFRASTRUCTURE.  Samples the bit counts of finished intra CUs (TEncCu::xCheckRDCostIntra :1483-1503) of two real encodes (64x64 golden
lenslet and the 64x64 sharp-edged frame) into tests/golden/encoder_intracu_calls.npz: the shim encoder (oracle/enc_shim.cpp; its bitstream equals the unmodified
reference's, tests/test_encoder_shim.py) runs with HOP_SHIM_TRACE_INTRACU; calls are kept spread over CU size, partition, tree depth, chroma direction and
transform skip: parameters, syntax elements, arrays, the CU's levels (Y | Cb | Cr), coder and CU-level contexts in and out, bits, distortion.  Replayed by
tests/test_oracle_golden5.py (restatement) and tests/test_gpu_tq_intra.py (hop_intra_cu_total_bits on the GPU).  Needs /root/reference (build container)."""
import os, struct, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import lenslet, sharp_frame
from make_golden9 import CFG, CODER
from make_golden12 import ISYN
PER_KIND = 2


def calls(frame):
    exe = os.path.join(ROOT, "oracle", "_ref", "TAppEncoderShim")
    Y, Cb, Cr = frame
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "in.yuv"), "wb").write(Y.astype(np.uint8).tobytes() + Cb.astype(np.uint8).tobytes() + Cr.astype(np.uint8).tobytes())
        tr = os.path.join(td, "t.bin")
        r = subprocess.run([exe, "-c", "/root/reference/cfg/3DHencoder_intra_main.cfg", "-i", "in.yuv", "-wdt", "64", "-hgt", "64", "-fr", "30", "-f", "1", "-q", "32",
                            "--MIsize=16", "-b", "s.bin", "-o", "rec.yuv"], cwd=td, capture_output=True, text=True, env=dict(os.environ, HOP_SHIM_TRACE_INTRACU=tr))
        assert r.returncode == 0, r.stderr[-2000:]
        b = open(tr, "rb").read()
    recs, o = [], 0
    while o < len(b):
        cfg = np.frombuffer(b, CFG, 1, o)[0]; o += CFG.itemsize
        syn = np.frombuffer(b, ISYN, 1, o)[0]; o += ISYN.itemsize
        arr = np.frombuffer(b, "u1", 1792, o).copy(); o += 1792
        cu = 1 << int(cfg["log2_cu"]); n = cu * cu * 3 // 2
        coef = np.frombuffer(b, "<i4", n, o).copy(); o += 4 * n
        cin = np.frombuffer(b, CODER, 1, o)[0]; o += 160
        cuin = np.frombuffer(b, "u1", 20, o).copy(); o += 20
        cout = np.frombuffer(b, CODER, 1, o)[0]; o += 160
        cuout = np.frombuffer(b, "u1", 20, o).copy(); o += 20
        bits, dist = struct.unpack_from("<2I", b, o); o += 8
        recs.append(dict(cfg=cfg, syn=syn, arr=arr, coef=coef, cin=cin, cuin=cuin, cout=cout, cuout=cuout, bits=bits, dist=dist))
    assert o == len(b)
    print(len(recs), "calls")
    return recs


def main():
    recs = calls(lenslet(64, 64, 16, 1234)) + calls(sharp_frame(64, 64, 77))
    rng = np.random.default_rng(17)
    groups = {}
    for r in recs:                                                      # (CU size, NxN, deepest transform depth, chroma is DM, transform skip, chroma levels)
        cu2 = 1 << (2 * int(r["cfg"]["log2_cu"])); parts = cu2 // 16
        key = (int(r["cfg"]["log2_cu"]), int(r["syn"]["part_nxn"]), int(r["arr"][:parts].max()), int(r["syn"]["chroma_is_dm"]), int(r["arr"][1024:].reshape(3, 256)[:, :parts].any()),
               int(r["coef"][cu2:].any()))
        groups.setdefault(key, []).append(r)
    keep = []
    for k in sorted(groups):
        L = groups[k]
        keep += [L[i] for i in rng.permutation(len(L))[:PER_KIND]]
    print(len(groups), "kinds ->", len(keep), "calls; sizes", sorted(set(int(r["cfg"]["log2_cu"]) for r in keep)))
    path = os.path.join(ROOT, "tests", "golden", "encoder_intracu_calls.npz")
    np.savez_compressed(path, cfg=np.array([r["cfg"] for r in keep]), syn=np.array([r["syn"] for r in keep]), arr=np.stack([r["arr"] for r in keep]),
                        coef=np.concatenate([r["coef"] for r in keep]), cin=np.array([r["cin"] for r in keep]), cuin=np.stack([r["cuin"] for r in keep]),
                        cout=np.array([r["cout"] for r in keep]), cuout=np.stack([r["cuout"] for r in keep]), bits=np.array([r["bits"] for r in keep], np.uint32),
                        dist=np.array([r["dist"] for r in keep], np.uint32))
    print(len(keep), "calls ->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
