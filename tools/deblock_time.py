#!/usr/bin/python3
"""tools/deblock_time.py -- hop_deblock_frame at the full frame size: a 7680x5376 picture tiled from a reference fixture (10 080 CTUs), kernel time from the HIP events the
library records around its five launches (hop_profile_*), checked against the CPU restatement.  Prints one JSON line."""
import ctypes, importlib.util, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import deblock_cases, oracle_deblock, tile_deblock_case
spec = importlib.util.spec_from_file_location("hophip", os.path.join(ROOT, "hevc-hop_amd", "hophip.py")); hp = importlib.util.module_from_spec(spec); spec.loader.exec_module(hp)
base = [c for c in deblock_cases() if (c[1], c[2]) == (256, 192)][0]
W, H, params, parts, pin = tile_deblock_case(base, 30, 28)
t0 = time.time(); want = oracle_deblock(W, H, params, parts, pin); cpu_s = time.time() - t0
ctx = hp.Context(W, H)
L = ctx.L
L.hop_profile_enable(ctx.h, 1)
reps = 5
for r in range(reps + 1):
    for c in range(3): ctx.plane_upload("recon", c, pin[c])
    if r == 1: L.hop_profile_reset(ctx.h)
    ctx.deblock_frame(parts, *params)
la, ms, un = ctypes.c_uint64(0), ctypes.c_double(0), ctypes.c_uint64(0)
L.hop_profile_read(ctx.h, 10, ctypes.byref(la), ctypes.byref(ms), ctypes.byref(un))
got = [ctx.recon_download(c) for c in range(3)]
ok = all(np.array_equal(a, b) for a, b in zip(got, want))
per = ms.value / max(1, la.value)
n_samples = W * H * 3 // 2
# algorithmic bytes of one call: each plane read and written once per direction (2 B samples), the partition data read twice per unit by the strength kernel, strengths written and read
alg = n_samples * 2 * 2 * 2 + (W // 4) * (H // 4) * (2 * 44 + 2 + 2)
print(json.dumps({"picture": "%dx%d" % (W, H), "ctus": (W // 64) * (H // 64), "equal_to_restatement": ok, "calls": int(la.value), "kernel_ms_per_call": per, "ctu_per_s_kernels": (W // 64) * (H // 64) / (per / 1e3),
                  "algorithmic_bytes": alg, "achieved_GBps": alg / (per / 1e3) / 1e9, "hbm_peak_GBps": 8000, "frac": alg / (per / 1e3) / 1e9 / 8000, "cpu_restatement_s": cpu_s, "cpu_ctu_per_s": (W // 64) * (H // 64) / cpu_s}))
ctx.close()
