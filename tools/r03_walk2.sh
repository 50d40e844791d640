#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_spine.py -x -q > gpurun_out/r03/t_walk2.log 2>&1; echo "spine: $(tail -n 1 gpurun_out/r03/t_walk2.log)"
HOP_PROF=1 python3 tools/enc_time.py 64 64 5 0 1 16 > gpurun_out/r03/walk_prof_1ctu.json 2> gpurun_out/r03/walk_prof_1ctu.err
HOP_PROF=1 python3 tools/enc_time.py 1280 128 5 0 1 16 > gpurun_out/r03/walk_prof_1280x128.json 2> gpurun_out/r03/walk_prof_1280x128.err
python3 tools/enc_time.py 1280 128 5 0 1 16 > gpurun_out/r03/walk2_1280x128.json 2>/dev/null
timeout -k 10 590 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > gpurun_out/r03/bench_b.json 2> gpurun_out/r03/bench_b.err; echo bench rc=$?
