#!/bin/bash
# the wavefront's lag against whole-frame parity: lag 8 and 6 (5 is the default so far)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
for lg in 8 6; do
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --views 0 --lag $lg > $O/bench_lag$lg.json 2> $O/bench_lag$lg.err || { echo "bench lag $lg failed"; tail -n 5 $O/bench_lag$lg.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/bench_lag$lg.json')); print('lag $lg value %.2f'%d['value'], d['steps'], d['parity']['ctus_compared'], d['parity']['mismatches'], d['parity']['mismatch_costs_here_reference'][:2], d['config']['rows_in_flight_max'], d['wavefront_visibility']['searches_reaching_above'], d['wavefront_visibility']['searches_reaching_below'])"
done
