#!/bin/bash
# lazy stash / restore+commit (now default) on the device, and the 256-register build of the walk kernels for launches of up to HOP_WALK_WIDE workgroups
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
HOP_WALK_WIDE=1024 timeout -k 10 560 python -m pytest tests/test_gpu_spine.py tests/test_gpu_tq_intra.py -x -q -k "spine or device_classes" > $O/t_n.log 2>&1 || { echo "tests FAILED"; tail -n 15 $O/t_n.log; exit 1; }
echo "spine + device classes (wide walks): $(tail -n 1 $O/t_n.log)"
for ww in 1024 0; do
  HOP_WALK_WIDE=$ww timeout -k 10 300 python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu --views 0 --budget-s 270 > $O/bench_n_$ww.json 2> $O/bench_n_$ww.err || { echo "bench $ww failed"; tail -n 5 $O/bench_n_$ww.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/bench_n_$ww.json')); print('walk_wide $ww value %.2f'%d['value'], d['steps'], d['parity']['mismatches'], {k:(round(v['ms']),v['calls']) for k,v in d['request_ms'].items() if v['calls']}, d['rendezvous'])"
done
