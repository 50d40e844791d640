#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
for k in 1 2 4 6; do
  HOP_SPINE_POSTED_KINDS=$k timeout -k 10 200 python -m pytest tests/test_gpu_spine.py -q -k "micro_image and 448 and 16" > $O/t_v_$k.log 2>&1; echo "kinds $k: $(tail -n 1 $O/t_v_$k.log)"
done
