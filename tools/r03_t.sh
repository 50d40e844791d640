#!/bin/bash
# stash / restore / commit posted by default, a flush at every CTU's end: the spine tests, then the bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
timeout -k 10 800 python -m pytest tests/test_gpu_spine.py tests/test_gpu_encoder_pic.py -x -q > $O/t_t.log 2>&1 || { echo "tests FAILED"; tail -n 15 $O/t_t.log; exit 1; }
echo "spine + encoder_pic: $(tail -n 1 $O/t_t.log)"
for po in 1 0; do
  HOP_SPINE_POSTED=$po timeout -k 10 200 python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu --views 0 --budget-s 170 > $O/bench_t_$po.json 2> $O/bench_t_$po.err || { echo "bench $po failed"; tail -n 5 $O/bench_t_$po.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/bench_t_$po.json')); print('posted $po value %.2f'%d['value'], d['steps'], d['parity']['mismatches'], {k:(round(v['ms']),v['calls']) for k,v in d['request_ms'].items() if v['calls']}, d['rendezvous'])"
done
