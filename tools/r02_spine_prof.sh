#!/bin/bash
# GPU spine tests, then rocprofv3 kernel trace of ONE CTU through hop_encode_frame (raster order): launches and time per kernel of the RD search
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_spine.py -x -q -s > $O/spine_gpu3.log 2>&1; tail -14 $O/spine_gpu3.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc1 -o enc1 -- python3 $R/tools/enc_time.py 64 64 > $O/enc1.json 2> $O/enc1.err || tail -5 $O/enc1.err
cat $O/enc1.json | cut -c1-300
HOP_GRAPHS=0 timeout -k 10 300 python3 $R/tools/enc_time.py 64 64 > $O/enc1_nograph.json 2>> $O/enc1.err; cut -c1-200 $O/enc1_nograph.json
