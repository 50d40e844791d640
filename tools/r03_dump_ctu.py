#!/usr/bin/env python3
"""tools/r03_dump_ctu.py -- DEVELOPER TOOL: code the bench frame on the device until CTU `a` is retired, then print the decisions of CTU a (per CU: position, size, mode,
partition, vectors) next to the reference's cost for it: where do the vectors of the first CTU that leaves the reference's costs point?"""
import importlib.util, json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import lenslet
spec = importlib.util.spec_from_file_location("hophip", os.path.join(ROOT, "hevc-hop_amd", "hophip.py")); hp = importlib.util.module_from_spec(spec); spec.loader.exec_module(hp)
a0 = int(sys.argv[1]) if len(sys.argv) > 1 else 3079
W, H = 7728, 5368
Y, Cb, Cr = lenslet(W, H, 15, 2)
g = np.load(os.path.join(ROOT, "tests", "golden", "encoder_frame_mi15_rows84.npz"))["cost"]
ctx = hp.Context(W, H, slots=48); ctx.upload_orig(Y, Cb, Cr)
out = {}
th = threading.Thread(target=lambda: out.update(r=ctx.encode_frame(32, 15, 0, '/tmp/r03_trace.txt', wpp=1, wavefront_lag=5))); th.start()
while th.is_alive():
    if ctx.encode_progress() >= a0 + 140: ctx.encode_cancel(); break
    time.sleep(0.01)
th.join()
cost, bits, dist, parts, nc = out["r"]
for a in (a0 - 1, a0, a0 + 1):
    print("CTU", a, "row", a // 121, "col", a % 121, "cost here", cost[a], "reference", g[a])
p = parts[a0]; x0, y0 = (a0 % 121) * 64, (a0 // 121) * 64
seen = set()
# z-order index -> position of the 4x4 unit
def zpos(i):
    x = y = 0
    for b in range(4): x |= ((i >> (2 * b)) & 1) << b; y |= ((i >> (2 * b + 1)) & 1) << b
    return x * 4, y * 4
for i in range(256):
    q = p[i]; d = int(q["depth"]); size = 64 >> d
    ux, uy = zpos(i); cx, cy = ux // size * size, uy // size * size
    key = (cx, cy, ux, uy) if int(q["part_size"]) else (cx, cy)
    pu = (int(q["mv"][0]), int(q["mv"][1]), int(q["merge_flag"]), int(q["gt_flag"]))
    k2 = (cx, cy, pu)
    if k2 in seen: continue
    seen.add(k2)
    print("  CU (%4d,%4d) %2d mode %d part %d skip %d | unit (%2d,%2d) mv %s (samples %.2f, %.2f) merge %d idx %d gt %d %s" % (x0 + cx, y0 + cy, size, int(q["pred_mode"]), int(q["part_size"]), int(q["skip"]), ux, uy,
          (int(q["mv"][0]), int(q["mv"][1])), int(q["mv"][0]) / 4.0, int(q["mv"][1]) / 4.0, int(q["merge_flag"]), int(q["merge_idx"]), int(q["gt_flag"]), [int(v) for v in q["gt"]] if int(q["gt_flag"]) else ""))
x0, y0 = (a0 % 121) * 64, (a0 // 121) * 64
print('candidate trace of the CTU (depth x y size pred_mode part_size skip merge bits dist cost):')
for ln in open('/tmp/r03_trace.txt'):
    t = ln.split()
    if len(t) > 3 and x0 <= int(t[1]) < x0 + 64 and y0 <= int(t[2]) < y0 + 64 and int(t[0]) <= 1: print('  ', ln.strip())
ctx.close()
