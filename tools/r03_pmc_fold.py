#!/usr/bin/env python3
"""tools/r03_pmc_fold.py SQ_DIR FETCH_DIR WRITE_DIR OUT.json -- fold the rocprofv3 counter passes of a small encode (tools/enc_time.py; separate --pmc passes, each with
--kernel-trace only) into per-launch figures of every kernel of the encode.  Units (MI355X_MICROARCH.md): SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles per
SIMD; FETCH_SIZE / WRITE_SIZE in KiB.  On gfx950 FETCH_SIZE reads half of a 16-B/lane streaming read; the encode's kernels read 2 - 8 B per lane (uncalibrated width), so the
raw value and the doubled one are both given: the truth lies between them."""
import csv, glob, json, sys, collections
def key(name): return name.replace("void ", "").split("(")[0].split("<")[0]
def fold(d):
    tot, n = collections.defaultdict(collections.Counter), collections.defaultdict(collections.Counter)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                k = key(r["Kernel_Name"]); tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    return tot, n
sq, sqn = fold(sys.argv[1]); ft, ftn = fold(sys.argv[2]); wt, wtn = fold(sys.argv[3])
out = {"_doc": __doc__.split(" -- ", 1)[1], "kernels": {}}
for k in sorted(set(sq) | set(ft) | set(wt)):
    e = {"launches": max([sqn[k][c] for c in sqn[k]] + [ftn[k]["FETCH_SIZE"], wtn[k]["WRITE_SIZE"], 0])}
    w = sq[k].get("SQ_WAVE_CYCLES", 0.0)
    for c, v in sq[k].items():
        e[c + "_per_launch"] = v / max(1, sqn[k][c])
        if w and c != "SQ_WAVE_CYCLES" and c != "SQ_WAVES": e[c + "_share_of_wave_cycles"] = v / w
    fb = ft[k]["FETCH_SIZE"] * 1024.0 / max(1, ftn[k]["FETCH_SIZE"]); wb = wt[k]["WRITE_SIZE"] * 1024.0 / max(1, wtn[k]["WRITE_SIZE"])
    e.update(fetch_bytes_per_launch_raw=fb, fetch_bytes_per_launch_doubled=2 * fb, write_bytes_per_launch=wb, hbm_bytes_per_launch=fb + wb, hbm_bytes_per_launch_upper=2 * fb + wb)
    out["kernels"][k] = e
json.dump(out, open(sys.argv[4], "w"), indent=1)
top = sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES_per_launch", 0) * kv[1]["launches"])[:12]
for k, e in top:
    print("%-28s n=%6d wait_any=%.2f valu=%.2f fetch=%.0f write=%.0f" % (k, e["launches"], e.get("SQ_WAIT_ANY_share_of_wave_cycles", -1), e.get("SQ_ACTIVE_INST_VALU_share_of_wave_cycles", -1),
                                                                        e["fetch_bytes_per_launch_raw"], e["write_bytes_per_launch"]))
