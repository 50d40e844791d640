#!/bin/bash
# final state of round 3: the whole GPU suite, the driver's bench command, two-rank rehearsals of the multi-GPU modes on the one GPU (gloo)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/t_final.log 2>&1 || { echo "GPU suite FAILED"; tail -n 25 $O/t_final.log; exit 1; }
echo "gpu suite: $(tail -n 1 $O/t_final.log)"
S=$(date +%s)
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$? wall=$(( $(date +%s) - S )) s"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_final.json'))
print('value %.2f'%d['value'], 'steps', d['steps'], d['timed_region'], d['parity']['mismatches'], d['cpu_baseline']['value'], d.get('cfg5_views'))
print({k:(round(v['ms']),v['calls']) for k,v in d['request_ms'].items() if v['calls']}, d['rendezvous'])
PY
S=$(date +%s)
timeout -k 10 300 python3 bench.py --gpus 2 --steps 4 --warmup 1 --rows 10 --no-cpu --views 0 --budget-s 200 --total-s 220 > $O/bench_final_2ranks.json 2> $O/bench_final_2ranks.err; echo "2 ranks rc=$? wall=$(( $(date +%s) - S )) s"; cut -c1-330 $O/bench_final_2ranks.json
timeout -k 10 300 python3 bench.py --gpus 2 --shard-rows --steps 4 --warmup 1 --rows 10 --no-cpu --views 0 --budget-s 200 --total-s 220 > $O/bench_final_shard2.json 2> $O/bench_final_shard2.err; echo "shard rc=$?"; cut -c1-330 $O/bench_final_shard2.json
