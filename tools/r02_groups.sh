#!/bin/bash
# picture groups (HOP_SPINE_GROUPS): parity of the stacked test, then timings at the bench's tile shape
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_spine.py -x -q -m gpu -k stacked > $O/gpu_groups.log 2>&1; rc=$?; echo "stacked rc $rc"; tail -3 $O/gpu_groups.log
[ $rc = 0 ] || exit 1
for cfg in "2 384" "4 384" "3 384"; do set -- $cfg
  HOP_SPINE_GROUPS=$1 timeout -k 10 300 python tools/enc_time.py 1024 256 5 0 $2 16 > $O/groups_$1_$2.json 2> $O/groups_$1_$2.err || { echo "G=$1 P=$2 failed"; tail -3 $O/groups_$1_$2.err; exit 1; }
  python - <<PY
import json; d=json.load(open("$O/groups_$1_$2.json")); print("G=$1 P=$2", round(d["ctu_per_s"],1), "CTU/s", round(d["s"],1), "s", {k:round(v["ms"]/1e3,1) for k,v in d["stats"].items() if "ms" in v}, d["stats"]["rendezvous"])
PY
done
