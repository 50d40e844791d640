#!/bin/bash
# round-3 baseline: latency of ONE CTU's chain and of a few wide single pictures, with and without posted requests
set -e
mkdir -p gpurun_out/r03
export HOP_SPINE_ROUND_STATS=1
python3 tools/enc_time.py 64 64 5 0 1 16 > gpurun_out/r03/base_1ctu.json 2> gpurun_out/r03/base_1ctu.err
HOP_SPINE_POSTED=1 python3 tools/enc_time.py 64 64 5 0 1 16 > gpurun_out/r03/base_1ctu_posted.json 2> gpurun_out/r03/base_1ctu_posted.err
HOP_GRAPHS=0 python3 tools/enc_time.py 64 64 5 0 1 16 > gpurun_out/r03/base_1ctu_nograph.json 2> gpurun_out/r03/base_1ctu_nograph.err
python3 tools/enc_time.py 1280 128 5 0 1 16 > gpurun_out/r03/base_1280x128.json 2> gpurun_out/r03/base_1280x128.err
HOP_SPINE_POSTED=1 python3 tools/enc_time.py 1280 128 5 0 1 16 > gpurun_out/r03/base_1280x128_posted.json 2> gpurun_out/r03/base_1280x128_posted.err
echo done
