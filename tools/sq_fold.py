#!/usr/bin/env python3
"""tools/sq_fold.py SQ_DIR FETCH_DIR WRITE_DIR STATS_CSV OUT.json -- fold the rocprofv3 passes of tools/r02_profile.sh (bench.py on ONE stream lane:
exclusive kernels) into one summary per hot kernel: SQ counters per launch, VALU-busy and LDS figures, HBM bytes per launch, average duration.
Counter units (MI355X_MICROARCH.md): SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES count quad-cycles per SIMD (x4 = cycles); SQ_LDS_IDX_ACTIVE = LDS-array cycles,
SQ_LDS_BANK_CONFLICT = the extra cycles conflicts cost; FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE reads half of a 16-B/lane streaming read on gfx950; these
kernels read 2-4 B per lane, uncalibrated: raw values kept)."""
import csv, glob, json, sys, collections
HOT = ("k_ss_family", "k_ss_search", "k_gt_search<unsigned short, 4, 64>", "k_gt_search<unsigned short, 1, 16>", "k_frac<4, 64>", "k_frac<1, 16>", "k_intra_rough",
       "k_pred_inter", "k_rdoq<5>", "k_rdoq<4>")
def key(name):
    return name.replace("void ", "").split("(")[0]
def fold(d):
    tot, n = collections.defaultdict(collections.Counter), collections.defaultdict(collections.Counter)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = key(r["Kernel_Name"])
            if k in HOT:
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    return tot, n
sq, sqn = fold(sys.argv[1]); ft, ftn = fold(sys.argv[2]); wt, wtn = fold(sys.argv[3])
dur = {}
for r in csv.DictReader(open(sys.argv[4])):
    k = key(r["Name"])
    if k in HOT: dur[k] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6}
N_SIMD, N_CU = 1024, 256
out = {"_doc": __doc__, "kernels": {}}
for k in HOT:
    if k not in sq or k not in dur: continue
    c, L = sq[k], max(1, sqn[k]["SQ_WAVE_CYCLES"])
    per = {a: b / L for a, b in c.items()}
    ms = dur[k]["avg_ms"]
    e = {"avg_launch_ms": ms, "launches_profiled": L, "sq_per_launch": {a: int(b) for a, b in per.items()},
         "lds_bank_conflict_over_idx_active": per["SQ_LDS_BANK_CONFLICT"] / max(1.0, per["SQ_LDS_IDX_ACTIVE"]),
         "valu_active_share_of_wave_cycles": per["SQ_ACTIVE_INST_VALU"] / max(1.0, per["SQ_WAVE_CYCLES"]),
         # cycles the 1024 SIMDs spent issuing VALU / (1024 SIMDs x duration x clock): the clock that makes this 1.0 is printed beside it
         "valu_simd_cycles": per["SQ_ACTIVE_INST_VALU"] * 4.0,
         "clock_ghz_at_which_valu_is_100pct_busy": per["SQ_ACTIVE_INST_VALU"] * 4.0 / N_SIMD / (ms * 1e-3) / 1e9,
         "valu_busy_at_2p1_ghz": per["SQ_ACTIVE_INST_VALU"] * 4.0 / N_SIMD / (ms * 1e-3 * 2.1e9),
         "lds_array_busy_at_2p1_ghz": per["SQ_LDS_IDX_ACTIVE"] / N_CU / (ms * 1e-3 * 2.1e9),
         "valu_insts_per_launch": per["SQ_INSTS_VALU"]}
    if k in ft: e["fetch_bytes_per_launch"] = ft[k]["FETCH_SIZE"] * 1024.0 / max(1, ftn[k]["FETCH_SIZE"])
    if k in wt: e["write_bytes_per_launch"] = wt[k]["WRITE_SIZE"] * 1024.0 / max(1, wtn[k]["WRITE_SIZE"])
    out["kernels"][k] = e
json.dump(out, open(sys.argv[5], "w"), indent=1)
for k, e in out["kernels"].items():
    print("%-40s %8.2f ms  valu@2.1GHz %.2f  lds@2.1GHz %.2f  conflict/idx %.2f  fetch %.2f GB write %.2f GB" % (k, e["avg_launch_ms"], e["valu_busy_at_2p1_ghz"], e["lds_array_busy_at_2p1_ghz"],
          e["lds_bank_conflict_over_idx_active"], e.get("fetch_bytes_per_launch", 0) / 1e9, e.get("write_bytes_per_launch", 0) / 1e9))
