#!/bin/bash
# the final predictions of a candidate travelling with its evaluation (HOP_SPINE_FUSE_PRED, default 1): the spine tests, then the bench with and without
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
timeout -k 10 560 python -m pytest tests/test_gpu_spine.py -x -q > $O/t_l.log 2>&1 || { echo "spine FAILED"; tail -n 15 $O/t_l.log; exit 1; }
echo "spine all: $(tail -n 1 $O/t_l.log)"
for fp in 1 0; do
  HOP_SPINE_FUSE_PRED=$fp timeout -k 10 300 python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu --views 0 --budget-s 270 > $O/bench_l_$fp.json 2> $O/bench_l_$fp.err || { echo "bench $fp failed"; tail -n 5 $O/bench_l_$fp.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/bench_l_$fp.json')); print('fuse_pred $fp value %.2f'%d['value'], d['steps'], d['parity']['mismatches'], {k:(round(v['ms']),v['calls']) for k,v in d['request_ms'].items() if v['calls']}, d['rendezvous'])"
done
