#!/bin/bash
# 48 candidate slots: a CU's first sub-CU evaluated with it.  The spine tests (0 / 16 / 24 / 48 slots), then the bench with 48 and 24 slots
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
timeout -k 10 800 python -m pytest tests/test_gpu_spine.py -x -q > $O/t_r.log 2>&1 || { echo "spine FAILED"; tail -n 15 $O/t_r.log; exit 1; }
echo "spine all: $(tail -n 1 $O/t_r.log)"
for sl in 48 24; do
  timeout -k 10 200 python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu --views 0 --budget-s 170 --slots $sl > $O/bench_r_$sl.json 2> $O/bench_r_$sl.err || { echo "bench $sl failed"; tail -n 5 $O/bench_r_$sl.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/bench_r_$sl.json')); print('slots $sl value %.2f'%d['value'], d['steps'], d['parity']['mismatches'], d['timed_region']['setup_s'], {k:(round(v['ms']),v['calls']) for k,v in d['request_ms'].items() if v['calls']}, d['rendezvous'])"
done
