#!/bin/bash
# 48 slots with 8 evaluation streams; stash / restore / commit posted on the device (HOP_SPINE_POSTED=1 HOP_SPINE_POSTED_DEVICE=1)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
run() { # name, env...
  n=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu --views 0 --budget-s 170 > $O/bench_s_$n.json 2> $O/bench_s_$n.err || { echo "bench $n failed"; tail -n 5 $O/bench_s_$n.err; return 1; }
  python3 -c "
import json; d=json.load(open('$O/bench_s_$n.json')); print('$n value %.2f'%d['value'], d['steps'], d['parity']['mismatches'], {k:(round(v['ms']),v['calls']) for k,v in d['request_ms'].items() if v['calls']}, d['rendezvous'])"
}
run streams8 HOP_SPINE_STREAMS=8 || exit 1
run streams4 HOP_SPINE_STREAMS=4 || exit 1
HOP_SPINE_STREAMS=8 HOP_SPINE_POSTED=1 HOP_SPINE_POSTED_DEVICE=1 timeout -k 10 400 python -m pytest tests/test_gpu_spine.py -x -q -k "448 or mi15 or bench_frame" > $O/t_s.log 2>&1 || { echo "posted spine FAILED"; tail -n 12 $O/t_s.log; exit 1; }
echo "posted spine subset: $(tail -n 1 $O/t_s.log)"
run posted8 HOP_SPINE_STREAMS=8 HOP_SPINE_POSTED=1 HOP_SPINE_POSTED_DEVICE=1 || exit 1
