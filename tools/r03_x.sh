#!/bin/bash
# where do the bench's parity mismatches come from: posted off / restore+commit / all three
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
run() { n=$1; shift
  env "$@" timeout -k 10 160 python3 bench.py --gpus 1 --steps 6 --warmup 1 --no-cpu --views 0 --budget-s 130 > $O/bench_x_$n.json 2> $O/bench_x_$n.err || { echo "bench $n failed"; tail -n 5 $O/bench_x_$n.err; return 1; }
  python3 -c "
import json; d=json.load(open('$O/bench_x_$n.json')); print('$n value %.2f'%d['value'], d['steps'], d['parity']['mismatches'], d['parity'].get('mismatch_ctus'), d['rendezvous']['rounds'])"
}
run posted0 HOP_SPINE_POSTED=0 || exit 1
run posted1 HOP_SPINE_POSTED=1 || exit 1
run posted1_t32 HOP_SPINE_POSTED=1 HOP_SPINE_THREADS=32 || exit 1
run posted0_again HOP_SPINE_POSTED=0 || exit 1
