#!/bin/bash
# one picture in WaveFrontSynchro mode, CTUs one after the other, through the CPU restatement and through the kernels, every request logged: where do they part?
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02; mkdir -p $O; cd $R
HOP_GRAPHS=0 timeout -k 10 600 python - <<'PY'
import os, sys, ctypes
sys.path.insert(0, "tests")
import numpy as np
from test_spine_cpu import *
import importlib.util
spec = importlib.util.spec_from_file_location("hophip", "hevc-hop_amd/hophip.py"); hp = importlib.util.module_from_spec(spec); spec.loader.exec_module(hp)
W, H, seed = 192, 128, 7
Y, Cb, Cr = frame(W, H, seed, False)
os.environ["HOP_SPINE_LOG"] = "gpurun_out/r02/log_cpu.bin"
run_cpu_wpp(spine_cpu(), W, H, Y, Cb, Cr, 0)
os.environ["HOP_SPINE_LOG"] = "gpurun_out/r02/log_gpu.bin"
ctx = hp.Context(W, H); ctx.upload_orig(Y, Cb, Cr)
ctx.encode_frame(32, 16, 0, None, wpp=1, wavefront_lag=1 << 20)
PY
python tools/spine_logdiff.py $O/log_cpu.bin $O/log_gpu.bin | tee $O/logdiff.txt
cp /tmp/spine_req.bin /tmp/spine_ans_a.bin /tmp/spine_ans_b.bin $O/ 2>/dev/null
rm -f $O/log_cpu.bin $O/log_gpu.bin
