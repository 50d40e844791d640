#!/bin/bash
# tools/r02_profile.sh -- one gpurun call: GPU tests, the default bench line, and the rocprofv3 passes profiles/r02_* are folded from.
# rocprofv3 gets the python program directly after `--` (no env/bash hop); counters are collected in their own passes with --kernel-trace only.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
timeout -k 10 600 python bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cut -c1-600 $O/bench.json
cd /tmp && export TMPDIR=/tmp
export HOP_LANES=1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/excl -o excl -- python3 $R/bench.py --steps 2 --warmup 0 --cpu-ctus 0 > $O/excl.json 2> $O/excl.err || { tail -20 $O/excl.err; exit 1; }
echo excl done
timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d $O/sq -o sq -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-ctus 0 > $O/sq.json 2> $O/sq.err || { tail -20 $O/sq.err; exit 1; }
echo sq done
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch -o fetch -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-ctus 0 > $O/fetch.json 2> $O/fetch.err || { tail -20 $O/fetch.err; exit 1; }
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write -o write -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-ctus 0 > $O/write.json 2> $O/write.err || { tail -20 $O/write.err; exit 1; }
echo pmc done
find $O -name "*.csv" | xargs ls -la | head -30
