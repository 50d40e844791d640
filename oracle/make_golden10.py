#!/usr/bin/env python3
"""oracle/make_golden10.py -- TEST INFRASTRUCTURE.  Samples the xAddSymbolBitsInter calls of two real encodes (the 128x128 golden lenslet and the
64x64 sharp-edged frame) into tests/golden/encoder_cubits_calls.npz: the shim encoder (oracle/enc_shim.cpp; its bitstream equals the unmodified
reference's, tests/test_encoder_shim.py) runs with HOP_SHIM_TRACE_CUBITS; per CU size 16 calls are kept, spread over partition shapes, merge /
skip cases and GT flags: parameters, syntax elements, transform depth / cbf / transform-skip arrays, levels, coder and CU-level context states in
and out, the bits and the skip decision.  Replayed by tests/test_oracle_golden5.py (restatement) and
tests/test_gpu_tq_intra.py::test_cu_bits_encoder_calls (hop_inter_cu_bits on the GPU).  Needs /root/reference (build container)."""
import os, struct, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import lenslet, sharp_frame
from make_golden9 import CFG, CODER
SYN = np.dtype([("part_size", "<i4"), ("n_pu", "<i4"), ("skip_flag", "<i4"), ("skip_ctx", "<i4"), ("amp_acc", "<i4"), ("is_min_cu", "<i4"), ("max_merge_cand", "<i4"),
                ("pu", [("merge_flag", "<i4"), ("merge_idx", "<i4"), ("mvd", "<i4", (2,)), ("mvp_idx", "<i4"), ("gt_flag", "<i4"), ("gt", "<i4", (8,))], (4,))])
PER_SIZE = 16


def calls(W, H, frame):
    exe = os.path.join(ROOT, "oracle", "_ref", "TAppEncoderShim")
    Y, Cb, Cr = frame
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "in.yuv"), "wb").write(Y.astype(np.uint8).tobytes() + Cb.astype(np.uint8).tobytes() + Cr.astype(np.uint8).tobytes())
        tr = os.path.join(td, "cub.bin")
        r = subprocess.run([exe, "-c", "/root/reference/cfg/3DHencoder_intra_main.cfg", "-i", "in.yuv", "-wdt", str(W), "-hgt", str(H), "-fr", "30", "-f", "1",
                            "-q", "32", "--MIsize=16", "-b", "s.bin", "-o", "rec.yuv"], cwd=td, capture_output=True, text=True, env=dict(os.environ, HOP_SHIM_TRACE_CUBITS=tr))
        assert r.returncode == 0, r.stderr[-2000:]
        b = open(tr, "rb").read()
    recs, o = [], 0
    while o < len(b):
        cfg = np.frombuffer(b, CFG, 1, o)[0]; o += CFG.itemsize
        syn = np.frombuffer(b, SYN, 1, o)[0]; o += SYN.itemsize
        arr = np.frombuffer(b, "u1", 256 * 7, o).copy(); o += 256 * 7
        cu = 1 << int(cfg["log2_cu"]); n = cu * cu * 3 // 2
        coef = np.frombuffer(b, "<i4", n, o).copy(); o += 4 * n
        cin = np.frombuffer(b, CODER, 1, o)[0]; o += 160
        cuin = np.frombuffer(b, "u1", 16, o).copy(); o += 16
        cout = np.frombuffer(b, CODER, 1, o)[0]; o += 160
        cuout = np.frombuffer(b, "u1", 16, o).copy(); o += 16
        bits, skipped = struct.unpack_from("<2i", b, o); o += 8
        recs.append(dict(cfg=cfg, syn=syn, arr=arr, coef=coef, cin=cin, cuin=cuin, cout=cout, cuout=cuout, bits=bits, skipped=skipped))
    print(len(recs), "calls")
    return recs


def main():
    assert SYN.itemsize == 7 * 4 + 4 * 14 * 4
    recs = calls(128, 128, lenslet(128, 128, 16, 1234)) + calls(64, 64, sharp_frame(64, 64, 77))
    rng = np.random.default_rng(10)
    keep = []
    for lg in (3, 4, 5, 6):
        L = [r for r in recs if int(r["cfg"]["log2_cu"]) == lg]
        groups = {}
        for r in L:                                                 # one bucket per (partition shape, skipped, any merge, any GT, root cbf)
            key = (int(r["syn"]["part_size"]), r["skipped"], int(r["syn"]["pu"]["merge_flag"].any()), int(r["syn"]["pu"]["gt_flag"].any()), int(r["arr"][256:1024:256].any()))
            groups.setdefault(key, []).append(r)
        order = sorted(groups)
        pick = []
        while len(pick) < PER_SIZE and any(groups.values()):
            for k in order:
                if groups[k] and len(pick) < PER_SIZE: pick.append(groups[k].pop(int(rng.integers(0, len(groups[k])))))
        print("size", 1 << lg, len(L), "calls in", len(order), "kinds ->", len(pick), "; partition shapes", sorted(set(int(r["syn"]["part_size"]) for r in pick)),
              "skipped", sum(r["skipped"] for r in pick), "GT", sum(int(r["syn"]["pu"]["gt_flag"].any()) for r in pick))
        keep += pick
    path = os.path.join(ROOT, "tests", "golden", "encoder_cubits_calls.npz")
    np.savez_compressed(path, cfg=np.array([r["cfg"] for r in keep]), syn=np.array([r["syn"] for r in keep]), arr=np.stack([r["arr"] for r in keep]),
                        coef=np.concatenate([r["coef"] for r in keep]), cin=np.array([r["cin"] for r in keep]), cuin=np.stack([r["cuin"] for r in keep]),
                        cout=np.array([r["cout"] for r in keep]), cuout=np.stack([r["cuout"] for r in keep]), bits=np.array([r["bits"] for r in keep], np.int32),
                        skipped=np.array([r["skipped"] for r in keep], np.int32))
    print(len(keep), "calls ->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
