#!/usr/bin/env python3
"""oracle/make_golden8.py -- TEST INFRASTRUCTURE.  Samples the xRateDistOptQuant calls of a real encode into
tests/golden/encoder_rdoq_calls.npz: the shim encoder (oracle/enc_shim.cpp; its bitstream equals the unmodified reference's,
tests/test_encoder_shim.py) runs the 128x128 golden lenslet with HOP_SHIM_TRACE_RDOQ; 25 calls per (size, luma/chroma, intra/SS)
bucket are kept with their own context-evolved bit-estimate table.  Replayed by tests/test_oracle_golden5.py (restatement) and
tests/test_gpu_tq_intra.py::test_rdoq_encoder_calls (hop_rdoq on the GPU).  Needs /root/reference (build container)."""
import os, struct, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import lenslet
W = H = 128; SEED = 1234; PER_BUCKET = 25


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", "TAppEncoderShim")
    Y, Cb, Cr = lenslet(W, H, 16, SEED)
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "in.yuv"), "wb").write(Y.astype(np.uint8).tobytes() + Cb.astype(np.uint8).tobytes() + Cr.astype(np.uint8).tobytes())
        tr = os.path.join(td, "rdoq.bin")
        r = subprocess.run([exe, "-c", "/root/reference/cfg/3DHencoder_intra_main.cfg", "-i", "in.yuv", "-wdt", str(W), "-hgt", str(H), "-fr", "30", "-f", "1",
                            "-q", "32", "--MIsize=16", "-b", "s.bin", "-o", "rec.yuv"], cwd=td, capture_output=True, text=True, env=dict(os.environ, HOP_SHIM_TRACE_RDOQ=tr))
        assert r.returncode == 0, r.stderr[-2000:]
        b = open(tr, "rb").read()
    recs, o = [], 0
    while o < len(b):
        hd = struct.unpack_from("<10i", b, o); o += 40
        lam = struct.unpack_from("<d", b, o)[0]; o += 8
        tab = np.frombuffer(b, "<i4", 244, o).copy(); o += 976
        n2 = 1 << (2 * hd[0])
        src = np.frombuffer(b, "<i4", n2, o).copy(); o += 4 * n2
        dst = np.frombuffer(b, "<i4", n2, o).copy(); o += 4 * n2
        recs.append((hd, lam, tab, src, dst))
    print(len(recs), "calls")
    rng = np.random.default_rng(8)
    buckets = {}
    for r in recs:
        buckets.setdefault((r[0][0], min(r[0][1], 1), r[0][2]), []).append(r)
    keep = []
    for k in sorted(buckets):
        L = buckets[k]
        nz = [r for r in L if r[0][9] > 0]                      # prefer calls that keep levels: the decisions are exercised
        pick = [nz[i] for i in rng.permutation(len(nz))[:PER_BUCKET - 5]] + [L[i] for i in rng.permutation(len(L))[:5]]
        keep += pick
        print(k, len(L), "calls,", len(nz), "with levels ->", len(pick))
    path = os.path.join(ROOT, "tests", "golden", "encoder_rdoq_calls.npz")
    np.savez_compressed(path, hd=np.array([r[0] for r in keep], np.int32), lam=np.array([r[1] for r in keep], np.float64),
                        tab=np.stack([r[2] for r in keep]), src=np.concatenate([r[3] for r in keep]), dst=np.concatenate([r[4] for r in keep]))
    print(len(keep), "calls ->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
