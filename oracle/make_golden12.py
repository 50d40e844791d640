#!/usr/bin/env python3
"""oracle/make_golden12.py -- TEST INFRASTRUCTURE.  Samples the xGetIntraBitsQT calls of two real encodes (64x64 golden lenslet and the 64x64 sharp-edged frame)
into tests/golden/encoder_intrabits_calls.npz: the shim encoder (oracle/enc_shim.cpp; its bitstream equals the unmodified reference's, tests/test_encoder_shim.py)
runs with HOP_SHIM_TRACE_INTRABITS; calls are kept spread over CU size, partition, node depth and luma / chroma: parameters, syntax elements, the node, transform
depth / cbf / transform-skip arrays, the levels of the current tree gathered from the layer buffers into the CU layout, coder and CU-level context states in and out,
the bits.  Replayed by tests/test_oracle_golden5.py (restatement) and tests/test_gpu_tq_intra.py::test_intra_cu_bits_encoder_calls (hop_intra_cu_bits on the GPU).
Needs /root/reference (build container)."""
import os, struct, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import lenslet, sharp_frame
from make_golden9 import CFG, CODER
ISYN = np.dtype([("part_nxn", "<i4"), ("skip_flag", "<i4"), ("skip_ctx", "<i4"), ("is_min_cu", "<i4"), ("luma_dir", "<i4", (4,)), ("preds", "<i4", (4, 3)), ("pred_num", "<i4", (4,)),
                 ("chroma_is_dm", "<i4"), ("chroma_dir", "<i4")])
PER_KIND = 3


def calls(frame):
    exe = os.path.join(ROOT, "oracle", "_ref", "TAppEncoderShim")
    Y, Cb, Cr = frame
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "in.yuv"), "wb").write(Y.astype(np.uint8).tobytes() + Cb.astype(np.uint8).tobytes() + Cr.astype(np.uint8).tobytes())
        tr = os.path.join(td, "ib.bin")
        r = subprocess.run([exe, "-c", "/root/reference/cfg/3DHencoder_intra_main.cfg", "-i", "in.yuv", "-wdt", "64", "-hgt", "64", "-fr", "30", "-f", "1", "-q", "32",
                            "--MIsize=16", "-b", "s.bin", "-o", "rec.yuv"], cwd=td, capture_output=True, text=True, env=dict(os.environ, HOP_SHIM_TRACE_INTRABITS=tr))
        assert r.returncode == 0, r.stderr[-2000:]
        b = open(tr, "rb").read()
    recs, o = [], 0
    while o < len(b):
        cfg = np.frombuffer(b, CFG, 1, o)[0]; o += CFG.itemsize
        syn = np.frombuffer(b, ISYN, 1, o)[0]; o += ISYN.itemsize
        nd = struct.unpack_from("<4i", b, o); o += 16
        arr = np.frombuffer(b, "u1", 256 * 7, o).copy().reshape(7, 256); o += 256 * 7
        lg = int(cfg["log2_cu"]); cu2 = 1 << (2 * lg); n = cu2 * 3 // 2
        layers = np.frombuffer(b, "<i4", 4 * n, o).reshape(4, n); o += 16 * n
        # the levels of the current tree in the CU layout: partition p from the layer of its transform depth (16 luma, 4 + 4 chroma levels per partition)
        coef = np.zeros(n, np.int32)
        for p in range(cu2 // 16):
            L = layers[int(cfg["log2_max_tu"]) - (lg - int(arr[0, p]))]
            coef[16 * p:16 * p + 16] = L[16 * p:16 * p + 16]
            coef[cu2 + 4 * p:cu2 + 4 * p + 4] = L[cu2 + 4 * p:cu2 + 4 * p + 4]
            coef[cu2 + cu2 // 4 + 4 * p:cu2 + cu2 // 4 + 4 * p + 4] = L[cu2 + cu2 // 4 + 4 * p:cu2 + cu2 // 4 + 4 * p + 4]
        cin = np.frombuffer(b, CODER, 1, o)[0]; o += 160
        cuin = np.frombuffer(b, "u1", 20, o).copy(); o += 20
        cout = np.frombuffer(b, CODER, 1, o)[0]; o += 160
        cuout = np.frombuffer(b, "u1", 20, o).copy(); o += 20
        bits = struct.unpack_from("<i", b, o)[0]; o += 4
        recs.append(dict(cfg=cfg, syn=syn, nd=nd, arr=arr.reshape(-1), coef=coef, cin=cin, cuin=cuin, cout=cout, cuout=cuout, bits=bits))
    print(len(recs), "calls")
    return recs


def main():
    assert ISYN.itemsize == 26 * 4
    recs = calls(lenslet(64, 64, 16, 1234)) + calls(sharp_frame(64, 64, 77))
    rng = np.random.default_rng(12)
    groups = {}
    for r in recs:                                                      # (CU size, NxN, node depth, luma, chroma, node has split below it, any level)
        parts = 1 << (2 * (int(r["cfg"]["log2_cu"]) - 2)); d = r["nd"][0]; np_ = parts >> (2 * d)
        sub = r["arr"][:256][r["nd"][1]:r["nd"][1] + np_]
        key = (int(r["cfg"]["log2_cu"]), int(r["syn"]["part_nxn"]), d, r["nd"][2], r["nd"][3], int(sub.max() > d), int(r["coef"].any()))
        groups.setdefault(key, []).append(r)
    keep = []
    for k in sorted(groups):
        L = groups[k]
        keep += [L[i] for i in rng.permutation(len(L))[:PER_KIND]]
    print(len(groups), "kinds ->", len(keep), "calls; sizes", sorted(set(int(r["cfg"]["log2_cu"]) for r in keep)))
    path = os.path.join(ROOT, "tests", "golden", "encoder_intrabits_calls.npz")
    np.savez_compressed(path, cfg=np.array([r["cfg"] for r in keep]), syn=np.array([r["syn"] for r in keep]), nd=np.array([r["nd"] for r in keep], np.int32),
                        arr=np.stack([r["arr"] for r in keep]), coef=np.concatenate([r["coef"] for r in keep]), cin=np.array([r["cin"] for r in keep]),
                        cuin=np.stack([r["cuin"] for r in keep]), cout=np.array([r["cout"] for r in keep]), cuout=np.stack([r["cuout"] for r in keep]),
                        bits=np.array([r["bits"] for r in keep], np.int32))
    print(len(keep), "calls ->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
