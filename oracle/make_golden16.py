#!/usr/bin/env python3
"""oracle/make_golden16.py -- TEST INFRASTRUCTURE.  Samples the estIntraPredChromaQT calls (chroma intra search of a CU) of two real encodes (64x64 golden lenslet and
the 64x64 sharp-edged frame) into tests/golden/encoder_csearch_calls.npz: the shim encoder (oracle/enc_shim.cpp; its bitstream equals the unmodified reference's,
tests/test_encoder_shim.py) runs with HOP_SHIM_TRACE_CSEARCH; calls are kept spread over CU size, partition, the luma tree's depth, the chosen direction and
transform skip: parameters, syntax elements (the luma directions decided before), neighbour flags of every node, the CU's chroma originals, the chroma
reconstruction pictures around the CU before the call, the arrays as the luma search left them, the CI_CURR_BEST coder; after: direction, distortion, arrays, the
CU's chroma levels, reconstruction planes and picture blocks.  Replayed by tests/test_oracle_golden5.py (restatement) and tests/test_gpu_tq_intra.py
(hop_intra_chroma_search on the GPU).  Needs /root/reference (build container)."""
import os, struct, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import lenslet, sharp_frame
from make_golden9 import CFG, CODER
from make_golden12 import ISYN
PER_KIND = 1
AV = 341 * 36


def calls(frame):
    exe = os.path.join(ROOT, "oracle", "_ref", "TAppEncoderShim")
    Y, Cb, Cr = frame
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "in.yuv"), "wb").write(Y.astype(np.uint8).tobytes() + Cb.astype(np.uint8).tobytes() + Cr.astype(np.uint8).tobytes())
        tr = os.path.join(td, "t.bin")
        r = subprocess.run([exe, "-c", "/root/reference/cfg/3DHencoder_intra_main.cfg", "-i", "in.yuv", "-wdt", "64", "-hgt", "64", "-fr", "30", "-f", "1", "-q", "32",
                            "--MIsize=16", "-b", "s.bin", "-o", "rec.yuv"], cwd=td, capture_output=True, text=True, env=dict(os.environ, HOP_SHIM_TRACE_CSEARCH=tr))
        assert r.returncode == 0, r.stderr[-2000:]
        b = open(tr, "rb").read()
    recs, o = [], 0
    while o < len(b):
        cfg = np.frombuffer(b, CFG, 1, o)[0]; o += CFG.itemsize
        syn = np.frombuffer(b, ISYN, 1, o)[0]; o += ISYN.itemsize
        nd = struct.unpack_from("<4i", b, o); o += 16
        avail = np.frombuffer(b, "u1", AV, o).copy(); o += AV
        cu = 1 << int(cfg["log2_cu"]); h2 = cu * cu // 4; W = cu + 1
        org = np.frombuffer(b, "<i2", 2 * h2, o).copy(); o += 4 * h2
        win = np.frombuffer(b, "<i2", 2 * W * W, o).copy(); o += 4 * W * W
        ain = np.frombuffer(b, "u1", 1792, o).copy(); o += 1792
        cin = np.frombuffer(b, CODER, 1, o)[0]; o += 160
        cuin = np.frombuffer(b, "u1", 20, o).copy(); o += 20
        mode, dist = struct.unpack_from("<2i", b, o); o += 8
        aout = np.frombuffer(b, "u1", 1792, o).copy(); o += 1792
        coef = np.frombuffer(b, "<i4", 2 * h2, o).copy(); o += 8 * h2
        reco = np.frombuffer(b, "<i2", 2 * h2, o).copy(); o += 4 * h2
        rec = np.frombuffer(b, "<i2", 2 * h2, o).copy(); o += 4 * h2
        recs.append(dict(cfg=cfg, syn=syn, nd=nd, avail=avail, org=org, win=win, ain=ain, cin=cin, cuin=cuin, mode=mode, dist=dist, aout=aout, coef=coef, reco=reco, rec=rec))
    assert o == len(b)
    print(len(recs), "calls")
    return recs


def main():
    recs = calls(lenslet(64, 64, 16, 1234)) + calls(sharp_frame(64, 64, 77))
    rng = np.random.default_rng(16)
    groups = {}
    for r in recs:                                                      # (CU size, NxN, deepest luma transform depth, chosen direction class, transform skip chosen, any level)
        parts = 1 << (2 * (int(r["cfg"]["log2_cu"]) - 2))
        key = (int(r["cfg"]["log2_cu"]), int(r["syn"]["part_nxn"]), int(r["ain"][:parts].max()), int(r["mode"]), int(r["aout"][1280:1280 + parts].any() or r["aout"][1536:1536 + parts].any()),
               int(r["coef"].any()))
        groups.setdefault(key, []).append(r)
    keep = []
    for k in sorted(groups):
        L = groups[k]
        keep += [L[i] for i in rng.permutation(len(L))[:PER_KIND]]

    print(len(groups), "kinds ->", len(keep), "calls; sizes", sorted(set(int(r["cfg"]["log2_cu"]) for r in keep)))
    path = os.path.join(ROOT, "tests", "golden", "encoder_csearch_calls.npz")
    cat = lambda k: np.concatenate([r[k] for r in keep])
    np.savez_compressed(path, cfg=np.array([r["cfg"] for r in keep]), syn=np.array([r["syn"] for r in keep]), nd=np.array([r["nd"] for r in keep], np.int32),
                        avail=np.stack([r["avail"] for r in keep]), org=cat("org"), win=cat("win"), ain=np.stack([r["ain"] for r in keep]), cin=np.array([r["cin"] for r in keep]),
                        cuin=np.stack([r["cuin"] for r in keep]), mode=np.array([r["mode"] for r in keep], np.int32), dist=np.array([r["dist"] for r in keep], np.uint32),
                        aout=np.stack([r["aout"] for r in keep]), coef=cat("coef"), reco=cat("reco"), rec=cat("rec"))
    print(len(keep), "calls ->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
