#!/usr/bin/env python3
"""tools/irqt_prof.py -- throughput of hop_intra_rqt_device (rest of row a8: the luma transform tree of an intra PU) per CU size (developer tool, GPU box).
One call = every third CU of a 7680x5376 frame in both directions (the PUs of a call must not lie in each other's neighbourhood); 2Nx2N PUs, the final pass
(bCheckFirst off) and the candidate pass (bCheckFirst on); then the whole luma search of a CU (hop_intra_luma_search_device).  Jobs, options, snapshots, results and levels stay in HBM."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
from scipy.ndimage import gaussian_filter
from bench import _hophip
hp = _hophip()
W, H = 7680, 5376
rng = np.random.default_rng(5)
f = gaussian_filter(rng.normal(0, 1, (H // 4, W // 4)).astype(np.float32), 2.0)
f = np.kron(f / np.abs(f).max(), np.ones((4, 4), np.float32))
Y = np.clip(128 + 90 * gaussian_filter(f, 1.5) + rng.normal(0, 2, (H, W)), 0, 255).astype(np.int16)
ctx = hp.Context(W, H)
C = np.ascontiguousarray(Y[::2, ::2] // 2 + 64)
ctx.upload_orig(Y, C, C)
snap = np.zeros((1, hp.CABAC_CTX_BYTES), np.uint8); cus = np.zeros((1, hp.CABAC_CU_CTX_BYTES), np.uint8)
ctx.L.hop_cabac_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]; ctx.L.hop_cabac_cu_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
ctx.L.hop_cabac_init(snap.ctypes.data, 3, 32); ctx.L.hop_cabac_cu_init(cus.ctypes.data, 3, 32)
LAM = 0.57 * 2.0 ** ((32 - 12) / 3.0)
dev = torch.device("cuda", 0)
ctx.L.hop_intra_rqt_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 8
print("CU   bCheckFirst    PUs  device s   kPU/s  Msamples/s   mean depth  cbf")
for lg in (3, 4, 5, 6):
    S = 1 << lg
    xs, ys = np.meshgrid(np.arange(S, W - 2 * S, 3 * S), np.arange(S, H - 2 * S, 3 * S))
    n = xs.size
    jobs = np.zeros(n, hp.RQT_JOB_DTYPE); syn = np.zeros(n, hp.INTRA_CU_SYNTAX_DTYPE); opts = np.zeros(n, hp.INTRA_RQT_OPT_DTYPE)
    jobs["x"], jobs["y"], jobs["log2_cu"], jobs["ctx_index"] = xs.ravel(), ys.ravel(), lg, 0
    jobs["qp_scaled"] = (32, 31, 31); jobs["sign_hide"] = 1; jobs["use_ts"] = 1; jobs["log2_max_tu"] = 5
    jobs["log2_min_tu_in_cu"] = {3: 2, 4: 2, 5: 3, 6: 4}[lg]
    jobs["lambda_rd"] = LAM; jobs["lambda_rdoq"] = (LAM, LAM, LAM); jobs["dist_weight"] = (1.0, 1.0)
    syn["is_min_cu"] = int(lg == 3); syn["luma_dir"] = rng.integers(0, 35, (n, 4)); syn["preds"] = (0, 1, 26); syn["pred_num"] = 3; syn["chroma_is_dm"] = 1; syn["b_luma"] = 1
    opts["strong"] = 1; opts["avail"] = (1 << 33) - 1
    dj = torch.from_numpy(jobs.view(np.uint8)).to(dev); dy = torch.from_numpy(syn.view(np.uint8)).to(dev); do = torch.from_numpy(opts.view(np.uint8)).to(dev)
    ds = torch.from_numpy(snap).to(dev); du = torch.from_numpy(cus).to(dev)
    dr = torch.zeros(n * hp.RQT_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev); dc = torch.zeros(n * S * S * 3 // 2, dtype=torch.int32, device=dev)
    for cf in (0, 1):
        for it in range(3):
            ctx.plane_upload("recon", 0, Y)                            # neighbours: the content itself
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx._chk(ctx.L.hop_intra_rqt_device(ctx.h, n, dj.data_ptr(), jobs[:1].ctypes.data, 0, cf, dy.data_ptr(), do.data_ptr(), ds.data_ptr(), du.data_ptr(), dr.data_ptr(),
                                                dc.data_ptr(), None, None), "intra_rqt_device")
            ctx.sync(); dt = time.perf_counter() - t0
        res = np.frombuffer(dr.cpu().numpy().tobytes(), hp.RQT_RESULT_DTYPE)
        print("%2dx%-2d %6d %10d %8.4f %8.1f %10.1f %10.2f  %3.0f%%" % (S, S, cf, n, dt, n / dt / 1e3, n * S * S / dt / 1e6,
              float(np.mean([r["tr_idx"][:S * S // 16].mean() for r in res[:2000]])), 100.0 * float(np.mean(res["cbf"][:, 0, 0] != 0))))
# the whole luma search of a CU (hop_intra_luma_search_device = estIntraPredQT): rough search, candidate list, 4..10 candidate trees, the final tree
print("CU   part     CUs  device s   kCU/s  Msamples/s   candidates")
keep_cls = []
ctx.L.hop_intra_luma_search_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 9
for lg, nxn in ((3, 1), (3, 0), (4, 0), (5, 0), (6, 0)):
    S = 1 << lg
    xs, ys = np.meshgrid(np.arange(S, W - 2 * S, 3 * S), np.arange(S, H - 2 * S, 3 * S))
    n = xs.size
    jobs = np.zeros(n, hp.RQT_JOB_DTYPE); syn = np.zeros(n, hp.INTRA_CU_SYNTAX_DTYPE); opts = np.zeros(n, hp.INTRA_RQT_OPT_DTYPE); sj = np.zeros(n, hp.INTRA_SEARCH_JOB_DTYPE)
    jobs["x"], jobs["y"], jobs["log2_cu"], jobs["ctx_index"] = xs.ravel(), ys.ravel(), lg, 0
    jobs["qp_scaled"] = (32, 31, 31); jobs["sign_hide"] = 1; jobs["use_ts"] = 1; jobs["log2_max_tu"] = 5
    jobs["log2_min_tu_in_cu"] = {3: 2, 4: 2, 5: 3, 6: 4}[lg]
    jobs["lambda_rd"] = LAM; jobs["lambda_rdoq"] = (LAM, LAM, LAM); jobs["dist_weight"] = (1.0, 1.0)
    syn["is_min_cu"] = int(lg == 3); syn["part_nxn"] = nxn; syn["chroma_is_dm"] = 1
    opts["strong"] = 1; opts["avail"] = (1 << 33) - 1
    nf = 8 if (S >> nxn) <= 8 else 3
    sj["left_dir"] = 1; sj["above_dir"] = 26; sj["rough_flags"] = 1; sj["sqrt_lambda"] = np.sqrt(LAM); sj["num_full_rd"] = nf
    dj = torch.from_numpy(jobs.view(np.uint8)).to(dev); dy = torch.from_numpy(syn.view(np.uint8)).to(dev); do = torch.from_numpy(opts.view(np.uint8)).to(dev)
    dsj = torch.from_numpy(sj.view(np.uint8)).to(dev); ds = torch.from_numpy(snap).to(dev); du = torch.from_numpy(cus).to(dev)
    dq = torch.zeros(n * hp.INTRA_SEARCH_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    dr = torch.zeros(n * hp.RQT_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev); dc = torch.zeros(n * S * S * 3 // 2, dtype=torch.int32, device=dev)
    dk = torch.zeros(n * S * S, dtype=torch.int16, device=dev)
    for it in range(3):
        ctx.plane_upload("recon", 0, Y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx._chk(ctx.L.hop_intra_luma_search_device(ctx.h, n, dj.data_ptr(), jobs[:1].ctypes.data, nxn, nf, dy.data_ptr(), do.data_ptr(), dsj.data_ptr(), ds.data_ptr(), du.data_ptr(),
                                                    dq.data_ptr(), dr.data_ptr(), dc.data_ptr(), dk.data_ptr()), "intra_luma_search_device")
        ctx.sync(); dt = time.perf_counter() - t0
    sr = np.frombuffer(dq.cpu().numpy().tobytes(), hp.INTRA_SEARCH_RESULT_DTYPE)
    print("%2dx%-2d %5s %8d %8.4f %8.1f %10.1f %10.2f" % (S, S, "NxN" if nxn else "2Nx2N", n, dt, n / dt / 1e3, n * S * S / dt / 1e6, float(sr["n_cand"][:, 0].mean())), end="")
    # the chroma search on the luma result (hop_intra_chroma_search_device = estIntraPredChromaQT), then the CU's bits and cost (hop_intra_cu_total_bits_device)
    syn2 = syn.copy(); syn2["luma_dir"] = sr["best_dir"]; syn2["preds"] = (0, 1, 26); syn2["pred_num"] = 3
    dy2 = torch.from_numpy(syn2.view(np.uint8)).to(dev)
    dcr = torch.zeros(n * 8, dtype=torch.uint8, device=dev); dk2 = torch.zeros(n * S * S // 2, dtype=torch.int16, device=dev)
    ctx.L.hop_intra_chroma_search_device.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 10
    for it in range(2):
        ctx.plane_upload("recon", 1, C); ctx.plane_upload("recon", 2, C)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx._chk(ctx.L.hop_intra_chroma_search_device(ctx.h, n, dj.data_ptr(), jobs[:1].ctypes.data, dy2.data_ptr(), do.data_ptr(), ds.data_ptr(), du.data_ptr(), dr.data_ptr(),
                                                      dcr.data_ptr(), dc.data_ptr(), dk2.data_ptr()), "intra_chroma_search_device")
        ctx.sync(); dtc = time.perf_counter() - t0
    ctx.L.hop_intra_cu_total_bits_device.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 12
    db = torch.zeros(n, dtype=torch.int32, device=dev); dco = torch.zeros(n, dtype=torch.float64, device=dev); dd = torch.zeros(n, dtype=torch.int32, device=dev)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx._chk(ctx.L.hop_intra_cu_total_bits_device(ctx.h, n, dj.data_ptr(), jobs[:1].ctypes.data, dy2.data_ptr(), dr.data_ptr(), dc.data_ptr(), dd.data_ptr(), ds.data_ptr(),
                                                      du.data_ptr(), db.data_ptr(), dco.data_ptr(), None, None), "intra_cu_total_bits_device")
        ctx.sync(); dtb = time.perf_counter() - t0
    print("   chroma %.4f s (%.1f kCU/s)   bits %.4f s   whole candidate %.1f kCU/s" % (dtc, n / dtc / 1e3, dtb, n / (dt + dtc + dtb) / 1e3))
    keep_cls.append((n, nxn, nf, jobs[:1].copy(), dj, dy, do, dsj, dq, dr, dcr, dc, dk, dk2, torch.zeros(n * hp.INTRA_CU_SYNTAX_DTYPE.itemsize, dtype=torch.uint8, device=dev), dd, db, dco, dt + dtc + dtb))
# all five classes of this phase of the frame at once (hop_intra_cu_device_classes: one chain per class, the classes on separate streams)
CLS = np.dtype([("n", "<i4"), ("part_nxn", "<i4"), ("num_full_rd", "<i4"), ("pad", "<i4"), ("cls", hp.RQT_JOB_DTYPE)] + [(k, "<u8") for k in
               ("d_jobs", "d_syntax", "d_opts", "d_sjobs", "d_sresults", "d_results", "d_cresults", "d_coef", "d_reco_y", "d_reco_c", "d_syntax_out", "d_dist", "d_bits", "d_cost",
                "d_ctx_out", "d_cu_ctx_out")])
descs = np.zeros(len(keep_cls), CLS)
for d, t in zip(descs, keep_cls):
    d["n"], d["part_nxn"], d["num_full_rd"], d["cls"] = t[0], t[1], t[2], t[3][0]
    for name, buf in zip(("d_jobs", "d_syntax", "d_opts", "d_sjobs", "d_sresults", "d_results", "d_cresults", "d_coef", "d_reco_y", "d_reco_c", "d_syntax_out", "d_dist", "d_bits", "d_cost"), t[4:18]):
        d[name] = buf.data_ptr()
ctx.L.hop_intra_cu_device_classes.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
# (the five sets overlap in the picture; the timing does not care, a caller would hand over sets that do not)
# calls 1-2 issue the launches one by one, call 3 captures them into a graph, calls 4-6 replay it
times = []
for it in range(6):
    ctx.plane_upload("recon", 0, Y); ctx.plane_upload("recon", 1, C); ctx.plane_upload("recon", 2, C)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx._chk(ctx.L.hop_intra_cu_device_classes(ctx.h, len(descs), descs.ctypes.data, ds.data_ptr(), du.data_ptr()), "hop_intra_cu_device_classes")
    ctx.sync(); dta = time.perf_counter() - t0; times.append(dta)
print("per call:", " ".join("%.4f" % t for t in times), "(launch by launch, launch by launch, capture, replay x3)")
print("all five classes at once: %.4f s (one after the other: %.4f s); %d candidates, %.1f kCU/s" % (dta, sum(t[18] for t in keep_cls), sum(t[0] for t in keep_cls), sum(t[0] for t in keep_cls) / dta / 1e3))
ctx.close()
