#!/bin/bash
# end-of-round measurements: the whole GPU suite, the default bench line, the rocprofv3 kernel summary of the same command at the profiled pass's breadth (graphs off)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests_final.log 2>&1; echo "tests rc $?"; tail -3 $O/gpu_tests_final.log
timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; cut -c1-330 $O/bench_default.json
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
cd /tmp && export TMPDIR=/tmp
HOP_GRAPHS=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_enc -o enc -- python3 $R/bench.py --pictures 16 --profile-pictures 16 --cpu-ctus 1 > $O/bench_rocprof16.json 2> $O/bench_rocprof16.err; echo "rocprof rc $?"
find /tmp/prof_enc -name "*kernel_stats*" -exec cp {} $O/r02_encode_kernel_stats16.csv \;
cut -c1-300 $O/bench_rocprof16.json
