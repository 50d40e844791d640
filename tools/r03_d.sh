#!/bin/bash
mkdir -p gpurun_out/r03
T='tests/test_gpu_spine.py::test_encode_frame_wavefront_equals_the_reference_with_wavefront_synchro[448-192-3-5-16]'
timeout -k 10 300 python -m pytest tests/test_gpu_tq_intra.py -x -q -k "device_classes" > gpurun_out/r03/t_d1.log 2>&1; echo "device_classes (cand parallel): $(tail -n 1 gpurun_out/r03/t_d1.log)"
timeout -k 10 200 python -m pytest "$T" -x -q > gpurun_out/r03/t_d2.log 2>&1; echo "448 wpp default: $(tail -n 1 gpurun_out/r03/t_d2.log)"
HOP_SPINE_POSTED=1 HOP_WALK=0 timeout -k 10 200 python -m pytest "$T" -x -q > gpurun_out/r03/t_d3.log 2>&1; echo "448 wpp posted nowalk: $(tail -n 1 gpurun_out/r03/t_d3.log)"
HOP_SPINE_POSTED=1 HOP_WALK_CAND=0 timeout -k 10 200 python -m pytest "$T" -x -q > gpurun_out/r03/t_d4.log 2>&1; echo "448 wpp posted walk serial-cand: $(tail -n 1 gpurun_out/r03/t_d4.log)"
HOP_PROF=1 python3 tools/enc_time.py 64 64 5 0 1 16 > gpurun_out/r03/cand_prof_1ctu.json 2> gpurun_out/r03/cand_prof_1ctu.err
python3 tools/enc_time.py 64 64 5 0 1 16 > gpurun_out/r03/cand_1ctu.json 2>/dev/null
python3 tools/enc_time.py 1280 128 5 0 12 16 > gpurun_out/r03/cand_12x1280.json 2>/dev/null
python3 - <<'PY'
import json
for f in ('cand_prof_1ctu','cand_1ctu','cand_12x1280'):
    d=json.load(open('gpurun_out/r03/%s.json'%f)); s=d['stats']
    print(f, 's %.2f'%d['s'], 'ctu/s %.1f'%d['ctu_per_s'], {k:(round(v['ms']),v['calls']) for k,v in s.items() if k in ('me_search','pred_inter','evaluation_wait','intra_cu','inter_cu')})
    for k,v in d.get('kernels',{}).items():
        if 'walk' in k: print('   ',k,v)
PY
timeout -k 10 500 python -m pytest tests/test_gpu_spine.py -x -q > gpurun_out/r03/t_d5.log 2>&1; echo "spine all: $(tail -n 1 gpurun_out/r03/t_d5.log)"
