#!/bin/bash
# the driver's bench command with the default lag of 8
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
S=$(date +%s)
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_final4.json 2> $O/bench_final4.err; echo "bench rc=$? wall=$(( $(date +%s) - S )) s"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_final4.json'))
print('value %.2f'%d['value'], 'steps', d['steps'], d['timed_region'], d['parity']['ctus_compared'], d['parity']['mismatches'], d['cpu_baseline']['value'], d.get('cfg5_views'))
print({k:(round(v['ms']),v['calls']) for k,v in d['request_ms'].items() if v['calls']}, d['rendezvous'], d['wavefront_visibility'])
print(d['config']['workload'][:400])
PY
