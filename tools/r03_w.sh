#!/bin/bash
# restore / commit posted (stash not), the chain of first sub-CUs with their parent: the whole GPU suite, then the bench at 48 and 96 slots
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/t_w.log 2>&1 || { echo "GPU suite FAILED"; tail -n 25 $O/t_w.log; exit 1; }
echo "gpu suite: $(tail -n 1 $O/t_w.log)"
for sl in 48 96; do
  timeout -k 10 200 python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu --views 0 --budget-s 170 --slots $sl > $O/bench_w_$sl.json 2> $O/bench_w_$sl.err || { echo "bench $sl failed"; tail -n 5 $O/bench_w_$sl.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/bench_w_$sl.json')); print('slots $sl value %.2f'%d['value'], d['steps'], d['parity']['mismatches'], d['timed_region']['setup_s'], {k:(round(v['ms']),v['calls']) for k,v in d['request_ms'].items() if v['calls']}, d['rendezvous'])"
done
