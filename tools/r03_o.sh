#!/bin/bash
# 24 candidate slots: the AMP shapes of a node with its first batch of candidates; lazy stash / restore+commit.  The spine tests (0 / 16 / 24 slots), then the bench with 24 and 16 slots
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
timeout -k 10 700 python -m pytest tests/test_gpu_spine.py -x -q > $O/t_o.log 2>&1 || { echo "spine FAILED"; tail -n 15 $O/t_o.log; exit 1; }
echo "spine all: $(tail -n 1 $O/t_o.log)"
for sl in 24 16; do
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu --views 0 --budget-s 270 --slots $sl > $O/bench_o_$sl.json 2> $O/bench_o_$sl.err || { echo "bench $sl failed"; tail -n 5 $O/bench_o_$sl.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/bench_o_$sl.json')); print('slots $sl value %.2f'%d['value'], d['steps'], d['parity']['mismatches'], {k:(round(v['ms']),v['calls']) for k,v in d['request_ms'].items() if v['calls']}, d['rendezvous'])"
done
