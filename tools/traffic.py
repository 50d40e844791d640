#!/usr/bin/env python3
"""tools/traffic.py FETCH_DIR WRITE_DIR OUT.json -- fold the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace
--output-format csv) of `python3 bench.py --steps 1 --warmup 0 --cpu-ctus 0` into HBM bytes per profiled launch region."""
import csv, glob, json, sys, collections
GROUP = (("k_gt_prep", "k_gt_search"), ("k_gt_search", "k_gt_search"), ("k_ss_", "k_ss_search"), ("k_frac", "k_frac"),
         ("k_pred_inter", "k_pred_inter"), ("k_ssref_commit", "k_ssref_commit"))
REGION_KERNEL = {"k_gt_search": "k_gt_search<unsigned short, 4", "k_ss_search": "k_ss_finalize", "k_frac": "k_frac<4", "k_pred_inter": "k_pred_inter", "k_ssref_commit": "k_ssref_commit"}
def fold(d, counter):
    tot, launches = collections.Counter(), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            name = r["Kernel_Name"].replace("void ", "")
            for pre, grp in GROUP:
                if name.startswith(pre):
                    tot[grp] += float(r["Counter_Value"]) * 1024.0      # counter unit: KiB
                    if name.startswith(REGION_KERNEL[grp]): launches[grp] += 1
                    break
    return tot, launches
ft, fl = fold(sys.argv[1], "FETCH_SIZE")
wt, wl = fold(sys.argv[2], "WRITE_SIZE")
out = {"_doc": "HBM traffic per kernel group from rocprofv3 PMC passes of `python3 bench.py --steps 1 --warmup 0 --cpu-ctus 0` on MI355X "
               "(separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes, each with --kernel-trace only). bytes = counter x 1024 (counter unit KiB). "
               "The gfx950 x2 correction of FETCH_SIZE applies to 16-B-per-lane streaming reads; these kernels read 2-4 B per lane "
               "(uncalibrated width), so the raw value is kept. per_launch = per profiled region of bench.py (one part of the frame's "
               "hop_me_search_device call: half of the 4.3 M PUs = 5082 CTUs with the default 2 stream lanes); k_gt_search covers k_gt_prep + both "
               "k_gt_search instantiations, k_ss_search covers the prep kernels + k_ss_family + k_ss_search + k_ss_finalize.", "kernels": {}}
for g in ("k_gt_search", "k_ss_search", "k_frac", "k_pred_inter", "k_ssref_commit"):
    n = max(1, fl[g])
    out["kernels"][g] = {"launches": n, "fetch_bytes_per_launch": ft[g] / n, "write_bytes_per_launch": wt[g] / max(1, wl[g]),
                         "hbm_bytes_per_launch": ft[g] / n + wt[g] / max(1, wl[g])}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
