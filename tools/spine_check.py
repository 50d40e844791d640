#!/usr/bin/env python3
"""tools/spine_check.py W H SEED [--sharp] -- build container only: run the reference encoder (oracle/_ref/TAppEncoderShim with its observers on) and the product's RD spine
over the CPU restatement (oracle/libhop_spine_cpu.so) on the same synthetic frame and compare candidate by candidate (the xCheckBestMode trace), CTU by CTU (cost.csv) and
the finished per-partition data.  Prints the first difference."""
import ctypes, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import lenslet, sharp_frame

def run_reference(W, H, seed, sharp, td, qp=32, mi=16, wpp=False, pitch=16):
    Y, Cb, Cr = sharp_frame(W, H, seed) if sharp else lenslet(W, H, pitch, seed)
    with open(os.path.join(td, "in.yuv"), "wb") as f:
        f.write(Y.astype(np.uint8).tobytes() + Cb.astype(np.uint8).tobytes() + Cr.astype(np.uint8).tobytes())
    env = dict(os.environ, HOP_SHIM_TRACE_BEST=os.path.join(td, "best.txt"), HOP_SHIM_TRACE_CTU=os.path.join(td, "ctu.bin"))
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "TAppEncoderShim"), "-c", "/root/reference/cfg/3DHencoder_intra_main.cfg", "-i", "in.yuv", "-wdt", str(W), "-hgt", str(H),
                        "-fr", "30", "-f", "1", "-q", str(qp), "--MIsize=%d" % mi, "-b", "s.bin", "-o", "rec.yuv"] +
                       (["--WaveFrontSynchro=1", "--WaveFrontSubstreams=%d" % ((H + 63) // 64)] if wpp else []), cwd=td, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return Y, Cb, Cr, time.time() - t0

def run_spine(W, H, Y, Cb, Cr, trace, qp=32, mi=16, first=0, wpp_lag=None):
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libhop_spine_cpu.so"))
    L.hop_spine_cpu_encode_wpp.restype = ctypes.c_long
    L.hop_spine_cpu_encode_wpp.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3 + [ctypes.c_char_p] + [ctypes.c_void_p] * 8
    L.hop_spine_cpu_encode.restype = ctypes.c_long
    L.hop_spine_cpu_encode.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3 + [ctypes.c_char_p] + [ctypes.c_void_p] * 8
    n = ((W + 63) // 64) * ((H + 63) // 64)
    cost = np.zeros(n, np.float64); bits = np.zeros(n, np.uint32); dist = np.zeros(n, np.uint32)
    ps = L.hop_spine_sizeof_part()
    parts = np.zeros((n * 256, ps), np.uint8)
    rec = [np.zeros((H, W), np.int16), np.zeros((H // 2, W // 2), np.int16), np.zeros((H // 2, W // 2), np.int16)]
    a = [np.ascontiguousarray(p, np.int16) for p in (Y, Cb, Cr)]
    t0 = time.time()
    if wpp_lag is not None:
        rr = np.zeros(2, np.float64)
        nc = L.hop_spine_cpu_encode_wpp(W, H, qp, mi, wpp_lag, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, trace.encode(), cost.ctypes.data, bits.ctypes.data, dist.ctypes.data,
                                        parts.ctypes.data, rec[0].ctypes.data, rec[1].ctypes.data, rec[2].ctypes.data, rr.ctypes.data)
        print("wavefront lag", wpp_lag, "rounds", int(rr[0]), "requests", int(rr[1]))
        return nc, cost, bits, dist, parts, rec, time.time() - t0
    nc = L.hop_spine_cpu_encode(W, H, qp, mi, first, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, trace.encode(), cost.ctypes.data, bits.ctypes.data, dist.ctypes.data,
                                parts.ctypes.data, rec[0].ctypes.data, rec[1].ctypes.data, rec[2].ctypes.data, None)
    return nc, cost, bits, dist, parts, rec, time.time() - t0

PART_DT = np.dtype([("depth", "u1"), ("pred_mode", "u1"), ("part_size", "u1"), ("skip", "u1"), ("merge_flag", "u1"), ("merge_idx", "u1"), ("gt_flag", "u1"), ("inter_dir", "u1"),
                    ("ref_idx", "i1"), ("mvp_idx", "i1"), ("mvp_num", "i1"), ("luma_dir", "u1"), ("chroma_dir", "u1"), ("tr_idx", "u1"), ("cbf", "u1", 3), ("tskip", "u1", 3),
                    ("mv", "i2", 2), ("mvd", "i2", 2), ("gt", "i2", 8)])

def read_ctu_trace(path):
    rec = np.dtype([("addr", "<i4"), ("cost", "<f8"), ("bits", "<u4"), ("dist", "<u4"), ("p", "<i2", (256, 23))])
    return np.fromfile(path, rec)

def main():
    W, H, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    sharp = "--sharp" in sys.argv
    wpp = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--wpp=")]      # --wpp=LAG: the reference with WaveFrontSynchro, the spine as a wavefront (0: serial)
    with tempfile.TemporaryDirectory() as td:
        Y, Cb, Cr, tr = run_reference(W, H, seed, sharp, td, wpp=bool(wpp))
        ref_lines = open(os.path.join(td, "best.txt")).read().split("\n")
        ctu = read_ctu_trace(os.path.join(td, "ctu.bin"))
        refrec = np.fromfile(os.path.join(td, "rec.yuv"), np.uint8)
        mine = os.path.join(td, "mine.txt")
        nc, cost, bits, dist, parts, rec, ts = run_spine(W, H, Y, Cb, Cr, mine, wpp_lag=wpp[0] if wpp else None)
        my_lines = open(mine).read().split("\n")
    print("reference %.1f s, %d candidates; spine %.1f s, %d candidates" % (tr, len(ref_lines) - 1, ts, nc))
    for i, (a, b) in enumerate(zip(ref_lines, my_lines)):
        if a != b:
            print("first difference at candidate %d:\n  ref : %s\n  mine: %s" % (i, a, b))
            for k in range(max(0, i - 6), i): print("  same:", ref_lines[k])
            return 1
    if len(ref_lines) != len(my_lines): print("trace lengths differ", len(ref_lines), len(my_lines)); return 1
    ok = True
    for c in ctu:
        a = int(c["addr"])
        if c["cost"] != cost[a] or c["bits"] != bits[a] or c["dist"] != dist[a]: print("CTU", a, "cost", c["cost"], cost[a], c["bits"], bits[a], c["dist"], dist[a]); ok = False
    P = parts.view(PART_DT).reshape(-1, 256)
    for c in ctu:
        a = int(c["addr"]); r = c["p"]; q = P[a]
        used = r[:, 1] != 15                                     # partitions outside the picture stay MODE_NONE
        pairs = (("depth", 0), ("pred_mode", 1), ("part_size", 2), ("skip", 3), ("merge_flag", 4), ("merge_idx", 5), ("gt_flag", 6), ("tr_idx", 9))
        for name, col in pairs:
            if not np.array_equal(q[name][used].astype(np.int16), r[used, col]): print("CTU", a, name, "differs"); ok = False
        intra = used & (r[:, 1] == 1)
        if not np.array_equal(q["luma_dir"][intra].astype(np.int16), r[intra, 7]) or not np.array_equal(q["chroma_dir"][intra].astype(np.int16), r[intra, 8]): print("CTU", a, "intra directions differ"); ok = False
        if not np.array_equal(q["cbf"][used].astype(np.int16), r[used, 10:13]): print("CTU", a, "cbf differs"); ok = False
        inter = used & (r[:, 1] == 0)
        if not np.array_equal(q["mv"][inter], r[inter, 13:15]) or not np.array_equal(q["gt"][inter], r[inter, 15:23]): print("CTU", a, "vectors differ"); ok = False
    print("per-CTU costs and partition data equal:", ok, list(cost[:6]))
    return 0 if ok else 1

if __name__ == "__main__":
    sys.exit(main())
