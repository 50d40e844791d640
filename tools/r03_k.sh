#!/bin/bash
# the walk kernels without the 128-register cap (libhophip_v256.so: __launch_bounds__(256, 2)): one CTU alone (per-launch times), then the bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
for v in std v256; do
  [ $v = v256 ] && export HOP_LIB=$R/hevc-hop_amd/libhophip_v256.so
  HOP_PROF=1 timeout -k 10 120 python3 tools/enc_time.py 64 64 5 0 1 16 > $O/k_prof_1ctu_$v.json 2>/dev/null || exit 1
  timeout -k 10 120 python3 tools/enc_time.py 64 64 5 0 1 16 > $O/k_1ctu_$v.json 2>/dev/null || exit 1
done
python3 - <<'PY'
import json
for f in ('k_prof_1ctu_std','k_prof_1ctu_v256','k_1ctu_std','k_1ctu_v256'):
    d=json.load(open('gpurun_out/r03/%s.json'%f)); s=d['stats']
    print(f, 's %.3f'%d['s'], {k:(round(v['ms']),v['calls']) for k,v in s.items() if k in ('me_search','pred_inter','evaluation_wait','intra_cu','inter_cu')})
    for k,v in d.get('kernels',{}).items():
        if 'walk' in k: print('   ',k,v)
PY
export HOP_LIB=$R/hevc-hop_amd/libhophip_v256.so
timeout -k 10 330 python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu --views 0 --budget-s 300 > $O/bench_k_v256.json 2> $O/bench_k_v256.err || { echo "bench failed"; tail -n 5 $O/bench_k_v256.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_k_v256.json')); print('v256 value %.2f'%d['value'], d['steps'], d['parity']['mismatches'], {k:(round(v['ms']),v['calls']) for k,v in d['request_ms'].items() if v['calls']}, d['rendezvous'])"
