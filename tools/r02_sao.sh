#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_sao.py tests/test_gpu_encoder_pic.py -x -q -m gpu -s -k "psnr or sao" > $O/gpu_sao.log 2>&1; rc=$?; echo "tests rc $rc"; tail -5 $O/gpu_sao.log
[ $rc = 0 ] && { timeout -k 10 200 python tools/sao_time.py > $O/sao_time.json 2> $O/sao_time.err; echo "time rc $?"; cat $O/sao_time.json; tail -2 $O/sao_time.err; }
