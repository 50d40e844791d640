#!/bin/bash
# the prediction-error request fused with the merge candidates' (spine tests), then the worker-thread count of the rendezvous at today's round structure
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_spine.py -x -q -k "24 or mi15 or bench_frame or two_contexts" > $O/t_p.log 2>&1 || { echo "spine FAILED"; tail -n 15 $O/t_p.log; exit 1; }
echo "spine subset: $(tail -n 1 $O/t_p.log)"
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
for th in 32 12 25; do
  HOP_SPINE_THREADS=$th timeout -k 10 240 python3 bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu --views 0 --budget-s 210 > $O/bench_p_$th.json 2> $O/bench_p_$th.err || { echo "bench $th failed"; tail -n 5 $O/bench_p_$th.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/bench_p_$th.json')); print('threads $th value %.2f'%d['value'], d['steps'], d['parity']['mismatches'], {k:(round(v['ms']),v['calls']) for k,v in d['request_ms'].items() if v['calls']}, d['rendezvous'])"
done
