#!/bin/bash
# One evidence set for profiles/: bench line, rocprofv3 kernel stats of the same command, PMC passes (HBM-side bytes,
# L2, TCP, TA), two-in-flight line.  Run on the GPU box:  bash tools/profile_round.sh <tag> [workload]
TAG=$1
WL=${2:-cfg3}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python3 bench.py --workload $WL --steps 3 --no-cpu-baseline 2> $O/${TAG}_pre.log > /dev/null   # page in
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o run -- python3 $R/bench.py --workload $WL --steps 3 --no-cpu-baseline > $O/${TAG}_stats_bench.json 2> $O/${TAG}_stats.log
f=$(find $O/${TAG}_stats -name "*kernel_stats.csv" | head -1)
cp $f $O/${TAG}_kernel_stats.csv
rm -rf $O/${TAG}_stats
# the same with every launch alone on the GPU (side stream off, one work set): each kernel's own duration, what bench.py's roofline divides by
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_solo_stats -o run -- python3 $R/bench.py --workload $WL --steps 3 --no-cpu-baseline --solo-steps > $O/${TAG}_solo_stats_bench.json 2> $O/${TAG}_solo_stats.log
f=$(find $O/${TAG}_solo_stats -name "*kernel_stats.csv" | head -1)
cp $f $O/${TAG}_solo_kernel_stats.csv
rm -rf $O/${TAG}_solo_stats
bash $R/tools/profile_pmc.sh $TAG --workload $WL
cp $O/pmc_$TAG/summary.json $O/${TAG}_pmc_summary.json
python3 $R/tools/pmc_family.py $O/${TAG}_pmc_summary.json $WL "profiles/${TAG}_pmc_summary.json (tools/profile_round.sh: separate rocprofv3 --pmc passes of bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --solo-steps)" > $O/${TAG}_pmc_${WL}.json
