#!/usr/bin/python3
"""tools/enc_time.py W H [lag [streams [pictures [slots]]]] -- time hop_encode_frame on a synthetic lenslet (pitch 15, seed 2 like bench.py): raster order, or the wavefront with batching."""
import importlib.util, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import lenslet
spec = importlib.util.spec_from_file_location("hophip", os.path.join(ROOT, "hevc-hop_amd", "hophip.py")); hp = importlib.util.module_from_spec(spec); spec.loader.exec_module(hp)
W, H = int(sys.argv[1]), int(sys.argv[2]); lag = int(sys.argv[3]) if len(sys.argv) > 3 else 0; streams = int(sys.argv[4]) if len(sys.argv) > 4 else 0; P = int(sys.argv[5]) if len(sys.argv) > 5 else 1; S = int(sys.argv[6]) if len(sys.argv) > 6 else 0
if P > 1:                                                    # a stack of P pictures: bands of one tall lenslet frame
    Yf, Cbf, Crf = lenslet(W, H * P, 15, 2)
    ctx = hp.Context(W, H, pictures=P, slots=S)
    ctx.upload_orig(ctx.stack([Yf[k * H:(k + 1) * H] for k in range(P)]), ctx.stack([Cbf[k * H // 2:(k + 1) * H // 2] for k in range(P)], True), ctx.stack([Crf[k * H // 2:(k + 1) * H // 2] for k in range(P)], True))
else:
    Y, Cb, Cr = lenslet(W, H, 15, 2)
    ctx = hp.Context(W, H, slots=S); ctx.upload_orig(Y, Cb, Cr)
prof = os.environ.get("HOP_PROF") == "1"            # HIP events around every launch (one stream): exclusive kernel times per kind
if prof:
    import ctypes
    L = ctx.L
    L.hop_profile_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.hop_profile_reset.argtypes = [ctypes.c_void_p]
    L.hop_profile_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.hop_profile_reset(ctx.h); L.hop_profile_enable(ctx.h, 1)
t0 = time.time()
cost, bits, dist, parts, nc = ctx.encode_frame(32, 15, 0, None, wpp=1 if lag else 0, wavefront_lag=lag, streams=streams)
dt = time.time() - t0
n = len(cost)
kinds = {}
if prof:
    names = ["ss_search", "frac", "gt_search", "pred", "commit", "dist", "leaf/tq", "intra", "rdoq", "cabac", "deblock", "sao", "walk_inter8", "walk_inter16", "walk_inter32", "walk_inter64",
             "walk_intra8", "walk_intra16", "walk_intra32", "walk_intra64", "walk_intra8_nxn"]
    for kid, name in enumerate(names):
        la, ms, un = ctypes.c_uint64(), ctypes.c_double(), ctypes.c_uint64()
        L.hop_profile_read(ctx.h, kid, ctypes.byref(la), ctypes.byref(ms), ctypes.byref(un))
        if la.value: kinds[name] = {"launches": la.value, "ms": round(ms.value, 2), "units": un.value, "us_per_launch": round(1e3 * ms.value / la.value, 1)}
print(json.dumps({"W": W, "H": H, "lag": lag, "streams": streams, "pictures": P, "slots": S, "ctus": n, "s": dt, "ctu_per_s": n / dt, "candidates": nc, "cost_sum": float(cost.sum()), "stats": ctx.encode_stats(), "kernels": kinds}))
