#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
bash tools/sweep.sh cfg3 64 1920 1080 134217728 > $O/r02_sweep_shade.txt 2>&1
echo "== default" >> $O/r02_sweep_shade.txt
python tools/quick_bench.py cfg3 64 1920 1080 134217728 2>&1 | tail -2 | head -1 >> $O/r02_sweep_shade.txt
cat $O/r02_sweep_shade.txt
