#!/bin/bash
# The gather micro-benchmark (tools/micro/gather_bench.hip): raw output at several table sizes, then the same binary
# under the L1 (TCP) counter pass that tools/profile_pmc.sh uses for bench.py, so that the traversal kernels' L1 access
# rate can be priced against the rate this loop reaches.  Run on the GPU box:  bash tools/profile_gather.sh <tag>
TAG=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
( for kb in 64 1024 8192 32768 102400 262144 1048576 2097152; do ./tools/micro/gather_bench $kb 2000; done ) > $O/${TAG}_gather_bench_raw.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for kb in 64 8192 102400 1048576; do
    timeout -k 5 300 rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $O/${TAG}_gather_pmc/kb$kb -- $R/tools/micro/gather_bench $kb 2000 > /dev/null 2> $O/${TAG}_gather_pmc_kb$kb.log || echo "gather pmc kb$kb failed"
done
# FETCH_SIZE calibration on an access pattern with a known byte count: 1 GB table, every record fetched from beyond L2 / Infinity Cache
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/${TAG}_gather_pmc/fetch_kb1048576 -- $R/tools/micro/gather_bench 1048576 2000 > /dev/null 2> $O/${TAG}_gather_fetch.log || echo "gather fetch pass failed"
rm -rf $O/${TAG}_gather_pmc/*/*/*kernel_trace* $O/${TAG}_gather_pmc/*/*/*agent_info*
python3 $R/tools/gather_json.py $O/${TAG}_gather_bench_raw.txt $O/${TAG}_gather_pmc > $O/${TAG}_gather_bench.json
