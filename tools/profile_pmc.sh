#!/bin/bash
# Collect PMC counters for bench.py in separate passes (FETCH_SIZE and WRITE_SIZE do not fit one TCC pass; the TA / TCP
# passes show the vector-memory path the traversal kernels are bound by).  Every pass is its own rocprofv3 run with
# --kernel-trace only, the program directly after `--`.  Run on the GPU box:  bash tools/profile_pmc.sh <tag> [bench args]
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {  # name counters...
    local name=$1; shift
    # a counter set the hardware cannot collect makes rocprofv3 abort and then hang: bound every pass
    timeout -k 5 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --solo-steps $BENCH_ARGS > /dev/null 2> $OUT/$name.log || echo "pass $name failed (see $name.log)"
}
BENCH_ARGS="$*"
pass sq SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY
pass fetch FETCH_SIZE TCC_HIT_sum
pass write WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum
pass tcp TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
pass ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY
pass td TD_TD_BUSY_sum TD_TC_STALL_sum
python3 $R/tools/pmc_summary.py $OUT/sq $OUT/sq2 $OUT/fetch $OUT/write $OUT/ea $OUT/tcp $OUT/tcp2 $OUT/ta $OUT/ta2 $OUT/grbm $OUT/td > $OUT/summary.json
rm -rf $OUT/*/*/*kernel_trace* $OUT/*/*/*agent_info*
du -sh $OUT
