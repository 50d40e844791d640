#!/bin/bash
# host side of the rendezvous at the breadth of one 7728-wide picture (24 CTU rows in flight: 12 stacked 1280x128 pictures): worker threads, spinning, posted requests, signal polling
mkdir -p gpurun_out/r03
run() { name=$1; shift; env "$@" python3 tools/enc_time.py 1280 128 5 0 12 16 > gpurun_out/r03/thr_$name.json 2> gpurun_out/r03/thr_$name.err; python3 - <<PY
import json
d=json.load(open('gpurun_out/r03/thr_$name.json'))
s=d['stats']; print('$name', 'ctu/s %.1f'%d['ctu_per_s'], 's %.1f'%d['s'], {k:(round(v['ms']),v['calls']) for k,v in s.items() if k in ('me_search','pred_inter','evaluation_wait','intra_cu','inter_cu','recon_stash')}, 'rounds', s['rendezvous']['rounds'], 'serve %.1f run %.1f'%(s['rendezvous']['serve_ms']/1e3, s['rendezvous']['run_ms']/1e3))
PY
}
run t24 HOP_X=1
run t8 HOP_SPINE_THREADS=8
run t4 HOP_SPINE_THREADS=4
run t2 HOP_SPINE_THREADS=2
run t1 HOP_SPINE_THREADS=1
run t4_spin0 HOP_SPINE_THREADS=4 HOP_SPINE_SPIN_US=0
run t4_posted HOP_SPINE_THREADS=4 HOP_SPINE_POSTED=1
run t4_poll HOP_SPINE_THREADS=4 HSA_ENABLE_INTERRUPT=0
run t4_posted_poll HOP_SPINE_THREADS=4 HOP_SPINE_POSTED=1 HSA_ENABLE_INTERRUPT=0
