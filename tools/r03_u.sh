#!/bin/bash
# which CTU: the mi15 pictures at 16 and 48 slots with stash / restore / commit posted (default) and not
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
for po in 1 0; do
  HOP_SPINE_POSTED=$po timeout -k 10 300 python -m pytest tests/test_gpu_spine.py -q -k "micro_image" > $O/t_u_$po.log 2>&1; echo "posted $po: $(tail -n 1 $O/t_u_$po.log)"
  grep -E "^E  |FAILED" $O/t_u_$po.log | head -12
done
HOP_SPINE_POSTED=0 timeout -k 10 300 python -m pytest tests/test_gpu_encoder_pic.py -q -k "448x192_seed3_mi15_wpp" > $O/t_u_pic0.log 2>&1; echo "pic posted 0: $(tail -n 1 $O/t_u_pic0.log)"
HOP_SPINE_POSTED=1 timeout -k 10 300 python -m pytest tests/test_gpu_encoder_pic.py -q -k "448x192_seed3_mi15_wpp" > $O/t_u_pic1.log 2>&1; echo "pic posted 1: $(tail -n 1 $O/t_u_pic1.log)"
