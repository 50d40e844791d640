#!/bin/bash
# end-of-round measurements: the whole GPU suite, the default bench line, smoke, the rocprofv3 kernel summary of the same command at the profiled pass's breadth (graphs
# off); a step runs only if the one before it ended normally
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02; mkdir -p $O; cd $R
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gpu_tests_final.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 $O/gpu_tests_final.log; [ $rc = 0 ] || exit 1
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; rc=$?; echo "bench rc $rc"; cut -c1-330 $O/bench_default.json; [ $rc = 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; rc=$?; echo "smoke rc $rc"; tail -3 $O/smoke.log; [ $rc = 0 ] || exit 1
cd /tmp && export TMPDIR=/tmp
HOP_GRAPHS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_enc -o enc -- python3 $R/bench.py --pictures 16 --profile-pictures 16 --cpu-ctus 1 > $O/bench_rocprof16.json 2> $O/bench_rocprof16.err; echo "rocprof rc $?"
find /tmp/prof_enc -name "*kernel_stats*" -exec cp {} $O/r02_encode_kernel_stats16.csv \;
cut -c1-300 $O/bench_rocprof16.json
