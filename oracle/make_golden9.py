#!/usr/bin/env python3
"""oracle/make_golden9.py -- TEST INFRASTRUCTURE.  Samples the xEstimateResidualQT calls of a real encode into
tests/golden/encoder_rqt_calls.npz: the shim encoder (oracle/enc_shim.cpp; its bitstream equals the unmodified reference's,
tests/test_encoder_shim.py) runs the 128x128 golden lenslet and a 64x64 sharp-edged frame (on which the 4x4 transform-skip variant wins)
with HOP_SHIM_TRACE_RQT; per CU size 14 + up to 6 calls are kept (the deepest transform trees first): parameters, coder state in / out, residual planes, cost / bits / distortion, the transform depth,
cbf and transform-skip arrays and the chosen levels, plus what encodeResAndCalcRdInterCU
made of it afterwards (HOP_SHIM_TRACE_FIN: reconstruction, original and the three final distortions).  Replayed by tests/test_oracle_golden5.py (restatement) and
tests/test_gpu_tq_intra.py::test_rqt_encoder_calls (hop_rqt on the GPU).  Needs /root/reference (build container)."""
import os, struct, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import lenslet, sharp_frame
W = H = 128; SEED = 1234; PER_SIZE = 14
CFG = np.dtype([("log2_cu", "<i4"), ("qp", "<i4", (3,)), ("bit_depth_y", "<i4"), ("bit_depth_c", "<i4"), ("sign_hide", "<i4"), ("use_ts", "<i4"), ("log2_max_tu", "<i4"),
                ("log2_min_tu_in_cu", "<i4"), ("inter_split_flag", "<i4"), ("pad", "<i4"), ("lambda_rd", "<f8"), ("lambda_rdoq", "<f8", (3,)), ("dist_weight", "<f8", (3,))])
CODER = np.dtype([("ctx", "u1", (150,)), ("pad", "u1", (2,)), ("frac", "<u8")])


def calls(W, H, frame):
    exe = os.path.join(ROOT, "oracle", "_ref", "TAppEncoderShim")
    Y, Cb, Cr = frame
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "in.yuv"), "wb").write(Y.astype(np.uint8).tobytes() + Cb.astype(np.uint8).tobytes() + Cr.astype(np.uint8).tobytes())
        tr = os.path.join(td, "rqt.bin"); tf = os.path.join(td, "fin.bin")
        r = subprocess.run([exe, "-c", "/root/reference/cfg/3DHencoder_intra_main.cfg", "-i", "in.yuv", "-wdt", str(W), "-hgt", str(H), "-fr", "30", "-f", "1",
                            "-q", "32", "--MIsize=16", "-b", "s.bin", "-o", "rec.yuv"], cwd=td, capture_output=True, text=True, env=dict(os.environ, HOP_SHIM_TRACE_RQT=tr, HOP_SHIM_TRACE_FIN=tf))
        assert r.returncode == 0, r.stderr[-2000:]
        b = open(tr, "rb").read(); fb = open(tf, "rb").read()
    recs, o = [], 0
    while o < len(b):
        cfg = np.frombuffer(b, CFG, 1, o)[0]; o += 104
        cin = np.frombuffer(b, CODER, 1, o)[0]; o += 160
        cout = np.frombuffer(b, CODER, 1, o)[0]; o += 160
        cu = 1 << int(cfg["log2_cu"]); n = cu * cu
        resi = np.frombuffer(b, "<i2", n * 3 // 2, o).copy(); o += n * 3
        cost = struct.unpack_from("<d", b, o)[0]; o += 8
        o4 = struct.unpack_from("<4I", b, o); o += 16
        arr = np.frombuffer(b, "u1", 256 * 7, o).copy(); o += 256 * 7
        fin = np.frombuffer(b, "<i4", n * 3 // 2, o).copy(); o += n * 6
        recs.append(dict(cfg=cfg, cin=cin, cout=cout, resi=resi, cost=cost, o4=o4, arr=arr, fin=fin))
    # the three final getDistPart calls of encodeResAndCalcRdInterCU after each quadtree (:6807-6810): reconstruction, original, distortion
    fo = 0
    for r in recs:
        cu = 1 << int(r["cfg"]["log2_cu"]); rec, org, d3 = [], [], []
        for k in range(3):
            w, dist = struct.unpack_from("<2i", fb, fo); fo += 8
            assert w == (cu >> 1 if k else cu), (w, cu, k)
            rec.append(np.frombuffer(fb, "<i2", w * w, fo).copy()); fo += 2 * w * w
            org.append(np.frombuffer(fb, "<i2", w * w, fo).copy()); fo += 2 * w * w
            d3.append(dist)
        r["rec"], r["org"], r["d3"] = np.concatenate(rec), np.concatenate(org), d3
    assert fo == len(fb)
    print(len(recs), "calls")
    return recs


def main():
    assert CFG.itemsize == 104 and CODER.itemsize == 160
    recs = calls(W, H, lenslet(W, H, 16, SEED))
    sharp = calls(64, 64, sharp_frame(64, 64, 77))                     # content on which the transform-skip variant wins (tests/hoputil.py)
    keep = []
    rng = np.random.default_rng(9)
    for lg in (3, 4, 5, 6):
        L = [r for r in recs if int(r["cfg"]["log2_cu"]) == lg]
        parts = 1 << (2 * (lg - 2))
        L.sort(key=lambda r: -(int(r["arr"][:parts].max()) * 1000 + int(np.count_nonzero(r["fin"]))))       # deepest trees, most levels first
        ts = [r for r in L if r["arr"][1024:].any()][:4]                  # calls in which the transform-skip variant won somewhere
        # calls after which the zero residual won the root test although the quadtree had levels (the reconstruction is the prediction)
        zr = [r for r in L if r["arr"][256:1024].any() and np.array_equal(r["rec"], (r["org"] - r["resi"]).astype(np.int16))][:3]
        pick = ts + zr + L[:PER_SIZE - 4 - len(ts) - len(zr)] + [L[i] for i in rng.permutation(len(L))[:4]]
        print("size", 1 << lg, len(L), "calls,", len(zr), "root-zero wins kept, max depth", [int(r["arr"][:parts].max()) for r in pick][:6], "transform skip", sum(int(r["arr"][1024:].any()) for r in pick))
        keep += pick
        S = [r for r in sharp if int(r["cfg"]["log2_cu"]) == lg and r["arr"][1024:].any()]
        keep += [S[i] for i in rng.permutation(len(S))[:6]]
        print("   + sharp frame:", len(S), "calls with transform skip chosen ->", min(6, len(S)))
    path = os.path.join(ROOT, "tests", "golden", "encoder_rqt_calls.npz")
    np.savez_compressed(path, cfg=np.array([r["cfg"] for r in keep]), cin=np.array([r["cin"] for r in keep]), cout=np.array([r["cout"] for r in keep]),
                        resi=np.concatenate([r["resi"] for r in keep]), cost=np.array([r["cost"] for r in keep]), o4=np.array([r["o4"] for r in keep], np.uint32),
                        arr=np.stack([r["arr"] for r in keep]), fin=np.concatenate([r["fin"] for r in keep]),
                        rec=np.concatenate([r["rec"] for r in keep]), org=np.concatenate([r["org"] for r in keep]), d3=np.array([r["d3"] for r in keep], np.uint32))
    print(len(keep), "calls ->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
