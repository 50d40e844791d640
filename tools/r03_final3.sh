#!/bin/bash
# one CTU alone with the final library (per-launch times of the walks, wall time), and the kernel-throughput mode of round 1 still runs
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
HOP_PROF=1 timeout -k 10 120 python3 tools/enc_time.py 64 64 5 0 1 48 > $O/final_prof_1ctu.json 2>/dev/null || exit 1
timeout -k 10 120 python3 tools/enc_time.py 64 64 5 0 1 48 > $O/final_1ctu.json 2>/dev/null || exit 1
python3 - <<'PY'
import json
for f in ('final_prof_1ctu','final_1ctu'):
    d=json.load(open('gpurun_out/r03/%s.json'%f)); s=d['stats']
    print(f, 's %.3f'%d['s'], {k:(round(v['ms']),v['calls']) for k,v in s.items() if k in ('me_search','pred_inter','evaluation_wait','intra_cu','inter_cu','recon_stash','commit')}, s['rendezvous'])
    for k,v in d.get('kernels',{}).items():
        if 'walk' in k: print('   ',k,v)
PY
timeout -k 10 300 python3 bench.py --kernels --steps 2 --warmup 1 > $O/final_kernels.json 2> $O/final_kernels.err; echo "kernels mode rc=$?"; cut -c1-300 $O/final_kernels.json
