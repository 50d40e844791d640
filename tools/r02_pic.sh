#!/bin/bash
# the picture-level binding on the GPU (reference encoder over hop_encode_frame -> the reference's bitstream), and the spine tests with the RD coder's fraction
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_encoder_pic.py -x -q -m gpu > $O/gpu_pic.log 2>&1; rc=$?; echo "pic rc $rc"; tail -5 $O/gpu_pic.log
[ $rc = 0 ] && { timeout -k 10 500 python -m pytest tests/test_gpu_spine.py -x -q -m gpu > $O/gpu_spine_frac.log 2>&1; echo "spine rc $?"; tail -3 $O/gpu_spine_frac.log; }
