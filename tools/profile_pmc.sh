#!/bin/bash
# Collect PMC counters for bench.py in separate passes (FETCH_SIZE and WRITE_SIZE do
# not fit one TCC pass).  Run on the GPU box:  bash tools/profile_pmc.sh <tag> [bench args]
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> $OUT/sq.log
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> $OUT/write.log
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/sq $OUT/fetch $OUT/write > $OUT/summary.json
rm -rf $OUT/sq/*/*kernel_trace* 
du -sh $OUT
