#!/usr/bin/env python3
"""oracle/make_golden11.py -- TEST INFRASTRUCTURE.  Records the xModeBitsIntra calls of a real encode (64x64 golden lenslet, shim encoder with
HOP_SHIM_TRACE_MODEBITS) and keeps 600 of them in tests/golden/encoder_modebits_calls.npz: context state, carried fraction, mode, most probable
modes -> bits.  Replayed by tests/test_oracle_golden5.py; the GPU's hop_intra_modes is tested against the restatement built from the same pieces.
Needs /root/reference (build container)."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import lenslet


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", "TAppEncoderShim")
    Y, Cb, Cr = lenslet(64, 64, 16, 1234)
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "in.yuv"), "wb").write(Y.astype(np.uint8).tobytes() + Cb.astype(np.uint8).tobytes() + Cr.astype(np.uint8).tobytes())
        tr = os.path.join(td, "mb.bin")
        r = subprocess.run([exe, "-c", "/root/reference/cfg/3DHencoder_intra_main.cfg", "-i", "in.yuv", "-wdt", "64", "-hgt", "64", "-fr", "30", "-f", "1", "-q", "32",
                            "--MIsize=16", "-b", "s.bin", "-o", "rec.yuv"], cwd=td, capture_output=True, text=True, env=dict(os.environ, HOP_SHIM_TRACE_MODEBITS=tr))
        assert r.returncode == 0, r.stderr[-2000:]
        rec = np.fromfile(tr, "<i4").reshape(-1, 8)
    print(len(rec), "calls; distinct (state, fraction):", len(np.unique(rec[:, :2], axis=0)))
    rng = np.random.default_rng(11)
    keep = rec[np.sort(rng.permutation(len(rec))[:600])]
    path = os.path.join(ROOT, "tests", "golden", "encoder_modebits_calls.npz")
    np.savez_compressed(path, rec=keep.astype(np.int32))
    print(len(keep), "calls ->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
