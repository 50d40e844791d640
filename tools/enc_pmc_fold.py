#!/usr/bin/env python3
"""tools/enc_pmc_fold.py SQ_DIR FETCH_DIR WRITE_DIR OUT.json -- fold the rocprofv3 counter passes of a small encode (bench.py --pictures 4, graphs off; separate --pmc
passes, each with --kernel-trace only) into per-launch figures of the encode's hot kernels.  Streams the counter CSVs (millions of rows).  Units (MI355X_MICROARCH.md):
SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles per SIMD; FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE reads half of a 16-B/lane streaming read on gfx950
-- these kernels read 2 - 8 B per lane (uncalibrated width): raw values are kept and said to be raw."""
import csv, glob, json, sys, collections
HOT = ("k_turd_fused_small", "k_turd_fused", "k_irqt_single<CabacLds1>", "k_rqt_single<CabacLds1>", "k_gt_search<unsigned short, 1, 16>", "k_gt_search<unsigned short, 4, 64>", "k_ss_search", "k_intra_modes")
def key(name): return name.replace("void ", "").split("(")[0]
def fold(d):
    tot, n = collections.defaultdict(collections.Counter), collections.defaultdict(collections.Counter)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                k = key(r["Kernel_Name"])
                if k in HOT: tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    return tot, n
sq, sqn = fold(sys.argv[1]); ft, ftn = fold(sys.argv[2]); wt, wtn = fold(sys.argv[3])
out = {"_doc": __doc__.split(" -- ", 1)[1], "kernels": {}}
for k in HOT:
    e = {}
    for c, v in sq[k].items(): e[c + "_per_launch"] = v / max(1, sqn[k][c])
    if "SQ_WAVE_CYCLES" in sq[k] and sq[k]["SQ_WAVE_CYCLES"]:
        w = sq[k]["SQ_WAVE_CYCLES"]
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in sq[k]: e[c + "_share_of_wave_cycles"] = sq[k][c] / w
    if ft[k]: e["fetch_bytes_per_launch_raw"] = ft[k]["FETCH_SIZE"] * 1024.0 / max(1, ftn[k]["FETCH_SIZE"])
    if wt[k]: e["write_bytes_per_launch"] = wt[k]["WRITE_SIZE"] * 1024.0 / max(1, wtn[k]["WRITE_SIZE"])
    e["launches"] = max([sqn[k][c] for c in sqn[k]] + [0])
    out["kernels"][k] = e
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out["kernels"].get("k_turd_fused_small", {}))[:600])
