#!/bin/bash
# the counting coder's tables in LDS (cb_tabs_load in every kernel that counts bins): the whole GPU suite, one CTU alone with per-launch times, the bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/t_m.log 2>&1 || { echo "GPU suite FAILED"; tail -n 25 $O/t_m.log; exit 1; }
echo "gpu suite: $(tail -n 1 $O/t_m.log)"
HOP_PROF=1 timeout -k 10 120 python3 tools/enc_time.py 64 64 5 0 1 16 > $O/m_prof_1ctu.json 2>/dev/null || exit 1
timeout -k 10 120 python3 tools/enc_time.py 64 64 5 0 1 16 > $O/m_1ctu.json 2>/dev/null || exit 1
python3 - <<'PY'
import json
for f in ('m_prof_1ctu','m_1ctu'):
    d=json.load(open('gpurun_out/r03/%s.json'%f)); s=d['stats']
    print(f, 's %.3f'%d['s'], {k:(round(v['ms']),v['calls']) for k,v in s.items() if k in ('me_search','pred_inter','evaluation_wait','intra_cu','inter_cu')})
    for k,v in d.get('kernels',{}).items():
        if 'walk' in k: print('   ',k,v)
PY
timeout -k 10 280 python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu --views 0 --budget-s 250 > $O/bench_m.json 2> $O/bench_m.err || { echo "bench failed"; tail -n 5 $O/bench_m.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_m.json')); print('value %.2f'%d['value'], d['steps'], d['parity']['mismatches'], {k:(round(v['ms']),v['calls']) for k,v in d['request_ms'].items() if v['calls']}, d['rendezvous'])"
