#!/bin/bash
# the whole GPU suite, the frame's top band as ONE picture (what a single picture's wavefront gives), and the rocprofv3 kernel summary at the profiled pass's breadth
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests_final.log 2>&1; echo "tests rc $?"; tail -3 $O/gpu_tests_final.log
timeout -k 10 420 python bench.py --tile-w 7728 --rows 8 --pictures 1 --profile-pictures 1 --cpu-ctus 1 > $O/bench_band8.json 2> $O/bench_band8.err; echo "band rc $?"; cut -c1-330 $O/bench_band8.json
cd /tmp && export TMPDIR=/tmp
HOP_GRAPHS=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_enc -o enc -- python3 $R/bench.py --pictures 16 --profile-pictures 16 --cpu-ctus 1 > $O/bench_rocprof16.json 2> $O/bench_rocprof16.err; echo "rocprof rc $?"
find /tmp/prof_enc -name "*kernel_stats*" -exec cp {} $O/r02_encode_kernel_stats16.csv \;
cut -c1-300 $O/bench_rocprof16.json
