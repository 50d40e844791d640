#!/bin/bash
# rocprofv3 kernel summaries of the stages after the search at frame size (tools/deblock_time.py, tools/sao_time.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dbk -o dbk -- python3 $R/tools/deblock_time.py > $O/deblock_time_rocprof.json 2> $O/deblock_time_rocprof.err; rc=$?; echo "deblock rc $rc"; [ $rc = 0 ] || exit 1
find /tmp/prof_dbk -name "*kernel_stats*" -exec cp {} $O/r02_deblock_kernel_stats.csv \;
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sao -o sao -- python3 $R/tools/sao_time.py > $O/sao_time_rocprof.json 2> $O/sao_time_rocprof.err; rc=$?; echo "sao rc $rc"
find /tmp/prof_sao -name "*kernel_stats*" -exec cp {} $O/r02_sao_kernel_stats.csv \;
head -8 $O/r02_deblock_kernel_stats.csv | cut -c1-160; head -6 $O/r02_sao_kernel_stats.csv | cut -c1-160
