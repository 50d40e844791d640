#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_gpu_tq_intra.py -x -q -k "device_classes" > gpurun_out/r03/t_e1.log 2>&1; echo "device_classes: $(tail -n 1 gpurun_out/r03/t_e1.log)"
timeout -k 10 590 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --views 0 > gpurun_out/r03/bench_e.json 2> gpurun_out/r03/bench_e.err; echo bench rc=$?
HOP_STREAM_PRIO=1 timeout -k 10 400 python3 bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu --views 0 > gpurun_out/r03/bench_e_prio.json 2> gpurun_out/r03/bench_e_prio.err; echo bench prio rc=$?
python3 - <<'PY'
import json
for f in ('bench_e','bench_e_prio'):
    d=json.load(open('gpurun_out/r03/%s.json'%f))
    print(f, 'value %.2f'%d['value'], 'steps', d['steps'], d['timed_region']['seconds'], d['parity'].get('mismatches'), {k:(round(v['ms']),v['calls']) for k,v in d['request_ms'].items() if v['calls']}, d['rendezvous']['run_ms'], d['rendezvous']['serve_ms'])
PY
