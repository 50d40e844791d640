#!/bin/bash
# how often does a run's parity check (row 0 of the frame against the reference's cost.csv) find a difference: eight short runs
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
for i in 1 2 3 4 5 6 7 8; do
  timeout -k 10 130 python3 bench.py --gpus 1 --steps 2 --warmup 0 --no-cpu --views 0 --budget-s 100 --profile-w 128 > $O/bench_y_$i.json 2> $O/bench_y_$i.err || { echo "bench $i failed"; tail -n 5 $O/bench_y_$i.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/bench_y_$i.json')); print('run $i value %.2f'%d['value'], d['steps'], 'mismatches', d['parity']['mismatches'], d['parity'].get('mismatch_ctus'), 'cost_sum', d['cost_sum_retired'], d['timed_region']['retired_total'])"
done
