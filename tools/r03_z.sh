#!/bin/bash
# the host's validity probe in raster semantics (nothing below the current CTU row counts as coded): does the frame's parity move?
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --views 0 > $O/bench_z.json 2> $O/bench_z.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_z.json'))
print('value %.2f'%d['value'], 'steps', d['steps'], d['parity']['ctus_compared'], d['parity']['mismatches'], d['parity']['mismatch_costs_here_reference'][:4], d['wavefront_visibility'])
PY
