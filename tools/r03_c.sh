#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 500 python -m pytest tests/test_gpu_spine.py -x -q -k "wavefront or micro_image or stacked" > gpurun_out/r03/t_c.log 2>&1; echo "spine subset: $(tail -n 1 gpurun_out/r03/t_c.log)"
HOP_SPINE_POSTED=1 timeout -k 10 400 python -m pytest tests/test_gpu_spine.py -x -q -k "wavefront or micro_image or stacked" > gpurun_out/r03/t_c_posted.log 2>&1; echo "spine subset posted: $(tail -n 1 gpurun_out/r03/t_c_posted.log)"
timeout -k 10 590 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --views 0 > gpurun_out/r03/bench_c.json 2> gpurun_out/r03/bench_c.err; echo bench rc=$?
