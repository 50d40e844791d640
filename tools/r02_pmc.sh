#!/bin/bash
# counter passes of a small encode (4 pictures in flight, graphs off): SQ occupancy / wait split and HBM traffic of the encode's hot kernels; only the fold comes back
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export HOP_GRAPHS=0
ARGS="--pictures 4 --profile-pictures 1 --cpu-ctus 1"
timeout -k 10 420 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES -d /tmp/pmc_sq -o sq -- python3 $R/bench.py $ARGS > $O/pmc_sq.json 2> $O/pmc_sq.err; echo "sq rc $?"
timeout -k 10 420 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pmc_fetch -o fetch -- python3 $R/bench.py $ARGS > $O/pmc_fetch.json 2> $O/pmc_fetch.err; echo "fetch rc $?"
timeout -k 10 420 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d /tmp/pmc_write -o write -- python3 $R/bench.py $ARGS > $O/pmc_write.json 2> $O/pmc_write.err; echo "write rc $?"
ls -la /tmp/pmc_sq/* | head -5
timeout -k 10 600 python3 $R/tools/enc_pmc_fold.py /tmp/pmc_sq /tmp/pmc_fetch /tmp/pmc_write $O/r02_encode_pmc.json
