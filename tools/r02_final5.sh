#!/bin/bash
# after the last change to the spine's worker pool: the spine tests again, then the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02; mkdir -p $O; cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_spine.py tests/test_gpu_encoder_pic.py -m gpu -x -q > $O/gpu_tests_final5.log 2>&1; rc=$?; echo "tests rc $rc"; tail -2 $O/gpu_tests_final5.log; [ $rc = 0 ] || exit 1
timeout -k 10 300 python bench.py > $O/bench_default5.json 2> $O/bench_default5.err; rc=$?; echo "bench rc $rc"; cut -c1-250 $O/bench_default5.json
