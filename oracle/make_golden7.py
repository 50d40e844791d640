#!/usr/bin/env python3
"""oracle/make_golden7.py -- TEST INFRASTRUCTURE.  Samples the calls a real encode makes into tests/golden/encoder_calls.npz.

Runs oracle/_ref/TAppEncoderShim (the reference encoder whose search members are oracle/enc_shim.cpp, see there; its bitstream is
checked against the unmodified reference's by tests/test_encoder_shim.py) on the 128x128 golden lenslet with HOP_SHIM_TRACE, joins
the xPatternSearch / xPatternSearchFracDIF / xPatternSearchGT records of one PU, and keeps a stratified sample: for each PU the job
as hop_me_search takes it, what the three members returned, and the SS-reference samples the members could read (everything else
of the plane is the -1 sentinel in the replay).  The replay tests put each PU back into a 128x128 context:
tests/test_oracle_golden5.py (restatement, CPU) and tests/test_gpu_parity.py::test_encoder_calls_replay (HIP path through the C ABI).
Needs /root/reference (build container)."""
import os, struct, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import lenslet
W = H = 128; SEED = 1234; PER_SHAPE = 3


def records(path):
    b = open(path, "rb").read(); o = 0
    while o < len(b):
        kind, w, h, nin, nout, x0, y0, ww, wh = struct.unpack_from("<9i", b, o); o += 36
        ins = np.frombuffer(b, "<i4", nin, o).copy(); o += 4 * nin
        outs = np.frombuffer(b, "<i8", nout, o).copy(); o += 8 * nout
        org = np.frombuffer(b, "<i2", w * h, o).reshape(h, w).copy(); o += 2 * w * h
        win = np.frombuffer(b, "<i2", ww * wh, o).reshape(wh, ww).copy(); o += 2 * ww * wh
        yield dict(kind=kind, w=w, h=h, ins=ins, outs=outs, org=org, win=win, x0=x0, y0=y0)


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", "TAppEncoderShim")
    Y, Cb, Cr = lenslet(W, H, 16, SEED)
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "in.yuv"), "wb").write(Y.astype(np.uint8).tobytes() + Cb.astype(np.uint8).tobytes() + Cr.astype(np.uint8).tobytes())
        tr = os.path.join(td, "trace.bin")
        r = subprocess.run([exe, "-c", "/root/reference/cfg/3DHencoder_intra_main.cfg", "-i", "in.yuv", "-wdt", str(W), "-hgt", str(H), "-fr", "30", "-f", "1",
                            "-q", "32", "--MIsize=16", "-b", "s.bin", "-o", "rec.yuv"], cwd=td, capture_output=True, text=True, env=dict(os.environ, HOP_SHIM_TRACE=tr))
        assert r.returncode == 0, r.stderr[-2000:]
        recs = list(records(tr))
    # join: an SS record followed by the frac and GT records of the same PU (xMotionEstimation calls them back to back, :4580-4632)
    pus = []
    for i in range(len(recs) - 2):
        a, b, c = recs[i], recs[i + 1], recs[i + 2]
        if (a["kind"], b["kind"], c["kind"]) == (0, 1, 2) and np.array_equal(a["org"], b["org"]) and np.array_equal(a["org"], c["org"]):
            pus.append((a, b, c))
    print(len(recs), "records,", len(pus), "complete PUs")
    rng = np.random.default_rng(7)
    by_shape = {}
    for p in pus:
        by_shape.setdefault((p[0]["w"], p[0]["h"]), []).append(p)
    keep = []
    for shape in sorted(by_shape):
        L = by_shape[shape]
        # prefer variety: one with the most sentinel samples in its window, one where the GT search changed something, the rest random
        L2 = sorted(L, key=lambda p: -int((p[0]["win"] == -1).sum()))[:1] + [p for p in L if p[2]["outs"][0] != 0][:1]
        idx = rng.permutation(len(L))[:PER_SHAPE]
        for p in L2 + [L[k] for k in idx]:
            if not any(p is q for q in keep): keep.append(p)
        keep = keep[:len(keep)]
    out = {}
    meta = []
    for n, (a, b, c) in enumerate(keep):
        px, py = int(c["ins"][14]), int(c["ins"][15])
        assert np.array_equal(Y[py:py + a["h"], px:px + a["w"]].astype(np.int16), a["org"]), "the PU's original block is the frame's"
        namvp = int(c["ins"][13])
        amvp = list(c["ins"][16:16 + 2 * namvp]) + [0] * (4 - 2 * namvp)
        #          0..3 rect        4..9 range + offsets    10..11 pred   12 lambda       13 fen  14 had          15 n_amvp 16..19 amvp
        meta.append([px, py, a["w"], a["h"]] + list(a["ins"][0:6]) + list(a["ins"][6:8]) + [int(a["ins"][8]), int(a["ins"][9]), int(c["ins"][12]), namvp] + amvp +
                    # 20..22 SS out        23..27 frac out       28..43 GT out
                    [int(v) for v in a["outs"]] + [int(v) for v in b["outs"]] + [int(v) for v in c["outs"]])
        assert list(b["ins"][0:2]) == [int(a["outs"][0]), int(a["outs"][1])] and list(c["ins"][0:2]) == list(b["ins"][0:2])
        for k, r in enumerate((a, b, c)):
            out["win%d_%d" % (n, k)] = r["win"]
            out["pos%d_%d" % (n, k)] = np.array([r["x0"], r["y0"]], np.int32)
    out["meta"] = np.array(meta, np.int64)
    out["W"] = np.int32(W); out["H"] = np.int32(H); out["seed"] = np.int32(SEED)
    path = os.path.join(ROOT, "tests", "golden", "encoder_calls.npz")
    np.savez_compressed(path, **out)
    print(len(keep), "PUs ->", path, os.path.getsize(path), "bytes; shapes:", sorted(by_shape))


if __name__ == "__main__":
    main()
