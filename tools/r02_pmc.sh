#!/bin/bash
# counter passes of a very small encode (one 128x128 picture = 4 CTUs, candidate slots on, graphs off): SQ occupancy / wait split and HBM traffic of the encode's hot
# kernels; only the fold comes back.  (The same passes over bench.py --pictures 4 did not finish in 7 minutes each: rocprofv3 serialises every dispatch when counters are on.)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export HOP_GRAPHS=0
ARGS="128 128 5 0 1 16"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES -d /tmp/pmc_sq -o sq -- python3 $R/tools/enc_time.py $ARGS > $O/pmc_sq.json 2> $O/pmc_sq.err; rc=$?; echo "sq rc $rc"
[ $rc = 0 ] && { timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pmc_fetch -o fetch -- python3 $R/tools/enc_time.py $ARGS > $O/pmc_fetch.json 2> $O/pmc_fetch.err; rc=$?; echo "fetch rc $rc"; }
[ $rc = 0 ] && { timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d /tmp/pmc_write -o write -- python3 $R/tools/enc_time.py $ARGS > $O/pmc_write.json 2> $O/pmc_write.err; rc=$?; echo "write rc $rc"; }
[ $rc = 0 ] && timeout -k 10 300 python3 $R/tools/enc_pmc_fold.py /tmp/pmc_sq /tmp/pmc_fetch /tmp/pmc_write $O/r02_encode_pmc.json
tail -2 $O/pmc_sq.json
