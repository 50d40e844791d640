#!/bin/bash
# tests of the walk kernels, the spine tests and the driver's bench command in one call (sequential: a failed step ends the call)
mkdir -p gpurun_out/r03
timeout -k 10 200 python -m pytest tests/test_gpu_tq_intra.py -x -q -k "device_classes" > gpurun_out/r03/t_g1.log 2>&1 || { echo "device_classes FAILED: $(tail -n 3 gpurun_out/r03/t_g1.log)"; exit 1; }
echo "device_classes: $(tail -n 1 gpurun_out/r03/t_g1.log)"
timeout -k 10 480 python -m pytest tests/test_gpu_spine.py -x -q > gpurun_out/r03/t_g2.log 2>&1 || { echo "spine FAILED: $(tail -n 3 gpurun_out/r03/t_g2.log)"; exit 1; }
echo "spine all: $(tail -n 1 gpurun_out/r03/t_g2.log)"
timeout -k 10 590 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench_g.json 2> gpurun_out/r03/bench_g.err; echo bench rc=$?
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_g.json'))
print('value %.2f'%d['value'], 'steps', d['steps'], d['timed_region'], d['parity'], d['cpu_baseline'], d.get('cfg5_views'))
print({k:(round(v['ms']),v['calls']) for k,v in d['request_ms'].items() if v['calls']}, d['rendezvous'])
PY
