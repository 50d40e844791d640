#!/usr/bin/env python3
"""tools/mt_bench.py -- do requests issued through several views of one context (own stream each, one host thread each) overlap on the device?  Each thread loops over
a small dependent chain (predictor launch + distortion launch + stream sync) on its own rectangle; variant 'host' uses the host-array entry (pageable copies)."""
import importlib.util, json, os, sys, threading, time, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from hoputil import lenslet
spec = importlib.util.spec_from_file_location("hophip", os.path.join(ROOT, "hevc-hop_amd", "hophip.py")); hp = importlib.util.module_from_spec(spec); spec.loader.exec_module(hp)
W, H = 2048, 512
Y, Cb, Cr = lenslet(W, H, 15, 2)
ctx = hp.Context(W, H); ctx.upload_orig(Y, Cb, Cr)
L = ctx.L
rects = np.array([[x, y, 64, 0] for y in range(0, H, 64) for x in range(0, W, 64)], np.int32)
L.hop_ssref_commit_cus(ctx.h, len(rects), rects.ctypes.data, np.ascontiguousarray(Y).ctypes.data, np.ascontiguousarray(Cb).ctypes.data, np.ascontiguousarray(Cr).ctypes.data)
L.hop_ctx_create_view.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
L.hop_sync.argtypes = [ctypes.c_void_p]
L.hop_pred_inter_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
L.hop_distortion_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
L.hop_distortion.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
K = int(os.environ.get("MT_ITERS", "2000"))
def worker(i, view, variant, out):
    pj = np.zeros(1, np.dtype([("pu_x", "<i4"), ("pu_y", "<i4"), ("w", "<i4"), ("h", "<i4"), ("mv_x", "<i4"), ("mv_y", "<i4"), ("use_gt", "<i4"), ("gt", "<i4", 8), ("dst_row_off", "<i4")]))
    pj["pu_x"], pj["pu_y"], pj["w"], pj["h"], pj["mv_x"], pj["mv_y"] = 256 + 64 * i, 256, 32, 32, -60, -4
    dj = np.array([(256 + 64 * i, 256, 32, 32, 0, 2)], np.dtype([("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4"), ("comp", "<i4"), ("kind", "<i4")]))
    d_pj = torch.from_numpy(pj.view(np.uint8)).cuda(); d_dj = torch.from_numpy(dj.view(np.uint8)).cuda(); d_out = torch.zeros(4, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    o = np.zeros(1, np.uint32)
    t0 = time.perf_counter()
    for _ in range(K):
        L.hop_pred_inter_device(view, 1, d_pj.data_ptr())
        if variant == "host": L.hop_distortion(view, 1, dj.ctypes.data, o.ctypes.data)
        else:
            L.hop_distortion_device(view, 1, d_dj.data_ptr(), d_out.data_ptr()); L.hop_sync(view)
    out[i] = time.perf_counter() - t0
res = {}
for variant in ("device", "host"):
    for nt in (1, 2, 4, 8, 16):
        views = []
        for i in range(nt):
            v = ctypes.c_void_p(); assert L.hop_ctx_create_view(ctx.h, ctypes.byref(v)) == 0; views.append(v)
        out = [0.0] * nt
        th = [threading.Thread(target=worker, args=(i, views[i], variant, out)) for i in range(nt)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        res["%s_%d" % (variant, nt)] = {"wall_s": dt, "chains_per_s": nt * K / dt, "us_per_chain_per_thread": dt / K * 1e6}
        for v in views: L.hop_ctx_destroy(v)
print(json.dumps({"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "iters": K, "results": res}))
