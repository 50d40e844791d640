#!/bin/bash
# the wave-per-unit leaves of the SS/GT candidate walk (HOP_WALK_WAVE_LEAVES=1): tests first, then timings with and without; every step only if the one before it passed
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
export HOP_WALK_WAVE_LEAVES=1
timeout -k 10 200 python -m pytest tests/test_gpu_tq_intra.py -x -q -k "inter_cu_device_classes" > $O/t_i1.log 2>&1 || { echo "device_classes FAILED"; tail -n 15 $O/t_i1.log; exit 1; }
echo "device_classes: $(tail -n 1 $O/t_i1.log)"
timeout -k 10 480 python -m pytest tests/test_gpu_spine.py -x -q -k "448 or mi15 or 200-136 or sharp" > $O/t_i2.log 2>&1 || { echo "spine FAILED"; tail -n 15 $O/t_i2.log; exit 1; }
echo "spine subset: $(tail -n 1 $O/t_i2.log)"
for wl in 0 1; do
  HOP_WALK_WAVE_LEAVES=$wl HOP_PROF=1 timeout -k 10 120 python3 tools/enc_time.py 64 64 5 0 1 16 > $O/i_prof_1ctu_$wl.json 2>/dev/null || exit 1
  HOP_WALK_WAVE_LEAVES=$wl timeout -k 10 120 python3 tools/enc_time.py 64 64 5 0 1 16 > $O/i_1ctu_$wl.json 2>/dev/null || exit 1
done
python3 - <<'PY'
import json
for f in ('i_prof_1ctu_0','i_prof_1ctu_1','i_1ctu_0','i_1ctu_1'):
    d=json.load(open('gpurun_out/r03/%s.json'%f)); s=d['stats']
    print(f, 's %.3f'%d['s'], {k:(round(v['ms']),v['calls']) for k,v in s.items() if k in ('me_search','pred_inter','evaluation_wait','intra_cu','inter_cu')})
    for k,v in d.get('kernels',{}).items():
        if 'walk_inter' in k: print('   ',k,v)
PY
for wl in 0 1; do
  HOP_WALK_WAVE_LEAVES=$wl timeout -k 10 330 python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu --views 0 --budget-s 300 > $O/bench_i_$wl.json 2> $O/bench_i_$wl.err || { echo "bench $wl failed"; tail -n 5 $O/bench_i_$wl.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/bench_i_$wl.json')); print('wave_leaves $wl value %.2f'%d['value'], d['steps'], d['parity']['mismatches'], {k:(round(v['ms']),v['calls']) for k,v in d['request_ms'].items() if v['calls']})"
done
