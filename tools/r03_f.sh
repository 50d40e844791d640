#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 200 python -m pytest tests/test_gpu_tq_intra.py -x -q -k "device_classes" > gpurun_out/r03/t_f1.log 2>&1; echo "device_classes: $(tail -n 1 gpurun_out/r03/t_f1.log)"
timeout -k 10 420 python -m pytest tests/test_gpu_spine.py -x -q > gpurun_out/r03/t_f2.log 2>&1; echo "spine all: $(tail -n 1 gpurun_out/r03/t_f2.log)"
HOP_PROF=1 python3 tools/enc_time.py 64 64 5 0 1 16 > gpurun_out/r03/f_prof_1ctu.json 2>/dev/null
python3 tools/enc_time.py 64 64 5 0 1 16 > gpurun_out/r03/f_1ctu.json 2>/dev/null
python3 - <<'PY'
import json
for f in ('f_prof_1ctu','f_1ctu'):
    d=json.load(open('gpurun_out/r03/%s.json'%f)); s=d['stats']
    print(f, 's %.2f'%d['s'], {k:(round(v['ms']),v['calls']) for k,v in s.items() if k in ('me_search','pred_inter','evaluation_wait','intra_cu','inter_cu')})
    for k,v in d.get('kernels',{}).items():
        if 'walk' in k: print('   ',k,v)
PY
timeout -k 10 590 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench_f.json 2> gpurun_out/r03/bench_f.err; echo bench rc=$?
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_f.json'))
print('value %.2f'%d['value'], 'steps', d['steps'], d['timed_region'], d['parity'], d['cpu_baseline'], d.get('cfg5_views'))
print({k:(round(v['ms']),v['calls']) for k,v in d['request_ms'].items() if v['calls']}, d['rendezvous'])
PY
