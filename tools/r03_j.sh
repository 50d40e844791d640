#!/bin/bash
# motion searches deferred behind the cheap requests of a node (HOP_SPINE_DEFER_ME=1): parity on the device, then the bench with and without
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
HOP_SPINE_DEFER_ME=1 timeout -k 10 400 python -m pytest tests/test_gpu_spine.py -x -q -k "448 or mi15 or bench_frame or two_contexts" > $O/t_j.log 2>&1 || { echo "spine FAILED"; tail -n 15 $O/t_j.log; exit 1; }
echo "spine subset (deferred searches): $(tail -n 1 $O/t_j.log)"
for dm in 1 0; do
  HOP_SPINE_DEFER_ME=$dm timeout -k 10 330 python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu --views 0 --budget-s 300 > $O/bench_j_$dm.json 2> $O/bench_j_$dm.err || { echo "bench $dm failed"; tail -n 5 $O/bench_j_$dm.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/bench_j_$dm.json')); print('defer_me $dm value %.2f'%d['value'], d['steps'], d['parity']['mismatches'], {k:(round(v['ms']),v['calls']) for k,v in d['request_ms'].items() if v['calls']}, d['rendezvous'])"
done
