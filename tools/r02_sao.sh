#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_sao.py tests/test_gpu_encoder_pic.py -x -q -m gpu -s > $O/gpu_sao.log 2>&1; rc=$?; echo "tests rc $rc"; tail -8 $O/gpu_sao.log
