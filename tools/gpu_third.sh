#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python3 tools/gather_json.py $O/r02_gather_bench_raw.txt $O/r02_gather_pmc > $O/r02_gather_bench.json
bash tools/sweep.sh cfg3 64 1920 1080 134217728 > $O/r02_sweep_block.txt 2>&1
echo "== default" >> $O/r02_sweep_block.txt
python tools/quick_bench.py cfg3 64 1920 1080 134217728 2>&1 | tail -2 | head -1 >> $O/r02_sweep_block.txt
cat $O/r02_sweep_block.txt
bash tools/gpu_bounces.sh
bash tools/profile_round.sh r02_a cfg3
ls $O | grep r02
