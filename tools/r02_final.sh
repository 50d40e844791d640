#!/bin/bash
# end-of-round measurements: the default bench line, a wider stack, and the rocprofv3 kernel summary of a smaller run of the same command (graphs off: the tracer and
# graph replays do not get along on this image); only the summaries come back
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02; mkdir -p $O; cd $R
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; cut -c1-400 $O/bench_default.json
cd /tmp && export TMPDIR=/tmp
HOP_GRAPHS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_enc -o enc -- python3 $R/bench.py --pictures 8 --profile-pictures 1 --cpu-ctus 1 > $O/bench_rocprof.json 2> $O/bench_rocprof.err; echo "rocprof rc $?"
find /tmp/prof_enc -name "*kernel_stats*" -exec cp {} $O/r02_encode_kernel_stats.csv \;
ls -la /tmp/prof_enc/* | head; cut -c1-300 $O/bench_rocprof.json
