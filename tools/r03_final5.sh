#!/bin/bash
# smoke() on the device, then the N = 2 path of the driver's bench command with both ranks on the one GPU (gloo)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -n 1 $O/smoke.log)"
S=$(date +%s)
timeout -k 10 500 python3 bench.py --gpus 2 --steps 4 --warmup 1 --budget-s 300 --total-s 420 > $O/bench_final5_2ranks.json 2> $O/bench_final5_2ranks.err; echo "2 ranks rc=$? wall=$(( $(date +%s) - S )) s"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_final5_2ranks.json'))
print('value %.2f'%d['value'], d['n_gpus'], d['steps'], d['scaling'], d['parity']['ctus_compared'], d['parity']['mismatches'], d.get('cfg5_views',{}).get('ctu_per_s_all_gpus'), d['config']['parallelism'])
PY
