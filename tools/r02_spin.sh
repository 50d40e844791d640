#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02; mkdir -p $O; cd $R
for cfg in "200 24" "0 32" "200 32"; do set -- $cfg
  HOP_SPINE_SPIN_US=$1 HOP_SPINE_THREADS=$2 timeout -k 10 200 python tools/enc_time.py 1024 256 5 0 384 16 > $O/spin_$1_$2.json 2> $O/spin_$1_$2.err || { echo "spin=$1 T=$2 failed"; tail -3 $O/spin_$1_$2.err; exit 1; }
  python - <<PY
import json; d=json.load(open("$O/spin_$1_$2.json")); rv=d["stats"]["rendezvous"]; print("spin_us=$1 T=$2", round(d["ctu_per_s"],1), "CTU/s", round(d["s"],1), "s serve", round(rv["serve_ms"]/1e3,1), "run", round(rv["run_ms"]/1e3,1))
PY
  grep -h "nr_throttled\|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
done
