#!/usr/bin/env python3
"""tools/r03_traffic.py [profiles/r03_encode_pmc.json] -> profiles/r03_traffic.json: HBM bytes per launch of every kernel of the encode from the counter passes
(tools/r03_pmc_fold.py), plus one synthesised entry: `k_intra_walk` = an intra candidate's whole chain (k_iw_begin, k_iw_cand and k_iw_pick per PU, k_iw_chroma, k_iw_finish;
what bench.py's profiled pass times as one region), bytes of all its kernels per chain."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_encode_pmc.json")
ks = json.load(open(src))["kernels"]
out = {"source": "profiles/r03_encode_pmc.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (each with --kernel-trace only) over `tools/enc_time.py 256 128 5 0 1 48` (8 CTUs of the "
                 "bench frame as one picture, lag-5 wavefront, 48 candidate slots); bytes = counter x 1024, averaged over the kernel's launches (all CU sizes of a walk kernel together). FETCH_SIZE "
                 "raw: the gfx950 x2 correction applies to 16-B/lane streaming reads, these kernels read 2-8 B per lane, so the true read volume lies between raw and 2x raw "
                 "(hbm_bytes_per_launch_upper).",
       "kernels": {k: {"launches": v["launches"], "hbm_bytes_per_launch": v["hbm_bytes_per_launch"], "hbm_bytes_per_launch_upper": v["hbm_bytes_per_launch_upper"],
                       "fetch_raw": v["fetch_bytes_per_launch_raw"], "write": v["write_bytes_per_launch"]} for k, v in ks.items() if v["launches"] > 0}}
iw = [k for k in out["kernels"] if k.startswith("k_iw_")]
if "k_iw_begin" in out["kernels"]:
    chains = out["kernels"]["k_iw_begin"]["launches"]
    tot = sum(out["kernels"][k]["hbm_bytes_per_launch"] * out["kernels"][k]["launches"] for k in iw); up = sum(out["kernels"][k]["hbm_bytes_per_launch_upper"] * out["kernels"][k]["launches"] for k in iw)
    out["kernels"]["k_intra_walk"] = {"launches": chains, "hbm_bytes_per_launch": tot / chains, "hbm_bytes_per_launch_upper": up / chains, "note": "per intra candidate chain: " + ", ".join(sorted(iw))}
json.dump(out, open(os.path.join(ROOT, "profiles", "r03_traffic.json"), "w"), indent=1)
print({k: round(v["hbm_bytes_per_launch"]) for k, v in out["kernels"].items() if "walk" in k or k.startswith("k_iw")})
