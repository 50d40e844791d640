#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_deblock.py tests/test_gpu_encoder_pic.py -x -q -m gpu -s > $O/gpu_deblock.log 2>&1; rc=$?; echo "tests rc $rc"; tail -6 $O/gpu_deblock.log
[ $rc = 0 ] && { timeout -k 10 200 python tools/deblock_time.py > $O/deblock_time.json 2> $O/deblock_time.err; echo "time rc $?"; cat $O/deblock_time.json; }
