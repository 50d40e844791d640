#!/usr/bin/python3
"""tools/sao_time.py -- hop_sao_stats / hop_sao_apply / hop_psnr at the full frame size (7680x5376 tiled from a reference fixture, 10 080 CTUs): kernel time from the HIP
events around the launches (hop_profile_*, HOP_K_SAO), the decision's host time, the CPU restatement beside them.  Prints one JSON line."""
import ctypes, importlib.util, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hoputil import SAO_PARAM_DTYPE, oracle, sao_cases
spec = importlib.util.spec_from_file_location("hophip", os.path.join(ROOT, "hevc-hop_amd", "hophip.py")); hp = importlib.util.module_from_spec(spec); spec.loader.exec_module(hp)
base = [c for c in sao_cases() if (c["W"], c["H"]) == (256, 192)][0]
nx, ny = 30, 28; W, H = 256 * nx, 192 * ny; n = (W // 64) * (H // 64)
org = [np.ascontiguousarray(np.tile(p, (ny, nx))) for p in base["org"]]; src = [np.ascontiguousarray(np.tile(p, (ny, nx))) for p in base["in"]]
ctx = hp.Context(W, H); L = ctx.L
ctx.upload_orig(*org)
for c in range(3): ctx.plane_upload("recon", c, src[c])
L.hop_profile_enable(ctx.h, 1)
def kernel_ms(fn, reps=4):
    fn(); L.hop_profile_reset(ctx.h)
    for _ in range(reps): fn()
    la, ms, un = ctypes.c_uint64(0), ctypes.c_double(0), ctypes.c_uint64(0)
    L.hop_profile_read(ctx.h, 11, ctypes.byref(la), ctypes.byref(ms), ctypes.byref(un))
    return ms.value / max(1, la.value)
stats_ms = kernel_ms(lambda: ctx.sao_stats())
stats = ctx.sao_stats()
p = hp.SaoParams((ctypes.c_double * 3)(*base["lambda"]), (ctypes.c_int32 * 3)(1, 1, 1), base["slice_type"], base["qp"], base["rd_fraction"])
coded = np.zeros((n, 3), SAO_PARAM_DTYPE); recon = np.zeros((n, 3), SAO_PARAM_DTYPE)
L.hop_sao_decide.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4
t0 = time.time(); assert L.hop_sao_decide(n, W // 64, 8, stats.ctypes.data, ctypes.addressof(p), coded.ctypes.data, recon.ctypes.data) == 0; decide_s = time.time() - t0
def apply():
    for c in range(3): ctx.plane_upload("recon", c, src[c])
    ctx.sao_apply(recon)
apply_ms = kernel_ms(apply)
t0 = time.time(); ssd, ps = ctx.psnr(); psnr_s = time.time() - t0
O = oracle()
ptr = lambda a: (ctypes.c_void_p * 3)(*[x.ctypes.data for x in a])
want = np.zeros_like(stats); t0 = time.time(); O.hop_o_sao_stats(W, H, 8, ptr(src), ptr(org), want.ctypes.data_as(ctypes.c_void_p)); cpu_stats_s = time.time() - t0
out = [np.zeros_like(a) for a in src]; t0 = time.time(); O.hop_o_sao_apply(W, H, 8, ptr(src), recon.ctypes.data_as(ctypes.c_void_p), ptr(out)); cpu_apply_s = time.time() - t0
ok = bool(np.array_equal(stats, want)) and all(np.array_equal(ctx.recon_download(c), out[c]) for c in range(3))
ns = W * H * 3 // 2
print(json.dumps({"picture": "%dx%d" % (W, H), "ctus": n, "equal_to_restatement": ok,
                  "stats_kernel_ms": stats_ms, "stats_algorithmic_bytes": ns * 4, "stats_GBps": ns * 4 / (stats_ms / 1e3) / 1e9, "stats_frac_of_8TBps": ns * 4 / (stats_ms / 1e3) / 8e12,
                  "apply_kernel_ms": apply_ms, "apply_algorithmic_bytes": ns * 4, "apply_GBps": ns * 4 / (apply_ms / 1e3) / 1e9, "apply_frac_of_8TBps": ns * 4 / (apply_ms / 1e3) / 8e12,
                  "decide_host_s": decide_s, "psnr_call_s": psnr_s, "cpu_restatement_stats_s": cpu_stats_s, "cpu_restatement_apply_s": cpu_apply_s}))
ctx.close()
