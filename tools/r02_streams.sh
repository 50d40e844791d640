#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02; mkdir -p $O; cd $R
for k in 4 8; do
  HOP_SPINE_STREAMS=$k timeout -k 10 200 python tools/enc_time.py 1024 256 5 0 384 16 > $O/streams_$k.json 2> $O/streams_$k.err || { echo "streams=$k failed"; tail -3 $O/streams_$k.err; exit 1; }
  python - <<PY
import json; d=json.load(open("$O/streams_$k.json")); s=d["stats"]; print("streams=$k", round(d["ctu_per_s"],1), "CTU/s", round(d["s"],1), "s; evaluation wait", round(s["evaluation_wait"]["ms"]/1e3,1), "me", round(s["me_search"]["ms"]/1e3,1), "pred", round(s["pred_inter"]["ms"]/1e3,1))
PY
done
