#!/bin/bash
# One evidence set for profiles/: bench line, rocprofv3 kernel stats of the same command, PMC passes,
# two-in-flight line.  Run on the GPU box:  bash tools/profile_round.sh <tag>
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python3 bench.py --steps 5 2> $O/${TAG}_bench.log | tail -1 > $O/${TAG}_bench.json
python3 bench.py --steps 5 --no-cpu-baseline --two-in-flight 2>> $O/${TAG}_bench.log | tail -1 > $O/${TAG}_two_in_flight.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o run -- python3 $R/bench.py --steps 3 --no-cpu-baseline > $O/${TAG}_stats_bench.json 2> $O/${TAG}_stats.log
f=$(find $O/${TAG}_stats -name "*kernel_stats.csv" | head -1)
cp $f $O/${TAG}_kernel_stats.csv
rm -rf $O/${TAG}_stats
bash $R/tools/profile_pmc.sh $TAG
cp $O/pmc_$TAG/summary.json $O/${TAG}_pmc_summary.json
