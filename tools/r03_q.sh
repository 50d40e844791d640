#!/bin/bash
# the whole GPU suite, then the driver's bench command
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/t_q.log 2>&1 || { echo "GPU suite FAILED"; tail -n 25 $O/t_q.log; exit 1; }
echo "gpu suite: $(tail -n 1 $O/t_q.log)"
S=$(date +%s)
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_q.json 2> $O/bench_q.err; echo "bench rc=$? wall=$(( $(date +%s) - S )) s"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_q.json'))
print('value %.2f'%d['value'], 'steps', d['steps'], d['timed_region'], d['parity'], d['cpu_baseline']['value'], d.get('cfg5_views'))
print({k:(round(v['ms']),v['calls']) for k,v in d['request_ms'].items() if v['calls']}, d['rendezvous'])
print(d['roofline'])
PY
