#!/bin/bash
# the driver's bench command with the visibility check, then rocprofv3 beside it (kernel trace + stats; the trace stays on the box) and three counter passes over a small encode
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
S=$(date +%s)
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_final2.json 2> $O/bench_final2.err; echo "bench rc=$? wall=$(( $(date +%s) - S )) s"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_final2.json'))
print('value %.2f'%d['value'], 'steps', d['steps'], d['parity']['ctus_compared'], d['parity']['mismatches'], d['parity']['mismatch_costs_here_reference'][:3], d['wavefront_visibility'], d.get('cfg5_views', {}).get('ctu_per_s_per_gpu'))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 640 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_rocprof2.json 2> $O/bench_rocprof2.err; rc=$?; echo "rocprof bench rc $rc"
[ $rc = 0 ] || { tail -n 20 $O/bench_rocprof2.err; exit 1; }
for f in $(find /tmp/prof_bench -name "*stats*.csv"); do cp $f $O/r03f_$(basename $f); done
ARGS="256 128 5 0 1 48"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES -d /tmp/pmc_sq -o sq -- python3 $R/tools/enc_time.py $ARGS > $O/pmc2_sq.json 2> $O/pmc2_sq.err; rc=$?; echo "sq rc $rc"
[ $rc = 0 ] || { tail -n 20 $O/pmc2_sq.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pmc_fetch -o fetch -- python3 $R/tools/enc_time.py $ARGS > $O/pmc2_fetch.json 2> $O/pmc2_fetch.err; rc=$?; echo "fetch rc $rc"
[ $rc = 0 ] || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d /tmp/pmc_write -o write -- python3 $R/tools/enc_time.py $ARGS > $O/pmc2_write.json 2> $O/pmc2_write.err; rc=$?; echo "write rc $rc"
[ $rc = 0 ] || exit 1
timeout -k 10 200 python3 $R/tools/r03_pmc_fold.py /tmp/pmc_sq /tmp/pmc_fetch /tmp/pmc_write $O/r03f_encode_pmc.json
