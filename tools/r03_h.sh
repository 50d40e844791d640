#!/bin/bash
# the sharded-picture test on the device, a two-rank rehearsal of bench.py --shard-rows (both ranks on the one GPU, gloo), then rocprofv3 beside the driver's bench command
# (kernel trace + stats; the trace itself stays on the box) and three counter passes over a small encode (each with --kernel-trace only)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_spine.py -x -q -k "two_contexts" > $O/t_h1.log 2>&1 || { echo "shard test FAILED"; tail -n 30 $O/t_h1.log; exit 1; }
echo "shard test: $(tail -n 1 $O/t_h1.log)"
timeout -k 10 300 python3 bench.py --gpus 2 --shard-rows --rows 10 --steps 3 --warmup 1 --budget-s 200 --total-s 220 --no-cpu --views 0 > $O/bench_shard2.json 2> $O/bench_shard2.err || { echo "shard bench FAILED"; tail -n 20 $O/bench_shard2.err; exit 1; }
cut -c1-400 $O/bench_shard2.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 640 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_rocprof.json 2> $O/bench_rocprof.err; rc=$?; echo "rocprof bench rc $rc"
[ $rc = 0 ] || { tail -n 20 $O/bench_rocprof.err; exit 1; }
for f in $(find /tmp/prof_bench -name "*stats*.csv"); do cp $f $O/r03_bench_$(basename $f); done
ls -la /tmp/prof_bench/* | head; cut -c1-300 $O/bench_rocprof.json
ARGS="256 128 5 0 1 16"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES -d /tmp/pmc_sq -o sq -- python3 $R/tools/enc_time.py $ARGS > $O/pmc_sq.json 2> $O/pmc_sq.err; rc=$?; echo "sq rc $rc"
[ $rc = 0 ] || { tail -n 20 $O/pmc_sq.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pmc_fetch -o fetch -- python3 $R/tools/enc_time.py $ARGS > $O/pmc_fetch.json 2> $O/pmc_fetch.err; rc=$?; echo "fetch rc $rc"
[ $rc = 0 ] || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d /tmp/pmc_write -o write -- python3 $R/tools/enc_time.py $ARGS > $O/pmc_write.json 2> $O/pmc_write.err; rc=$?; echo "write rc $rc"
[ $rc = 0 ] || exit 1
timeout -k 10 200 python3 $R/tools/r03_pmc_fold.py /tmp/pmc_sq /tmp/pmc_fetch /tmp/pmc_write $O/r03_encode_pmc.json
