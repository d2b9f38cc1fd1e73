#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
for m in mixed lambert; do echo "== $m"; YK_DEBUG_BOUNCES=1 python3 tools/matdiv_experiment.py $m 2>&1 | grep "^bounce" | tail -8; done > $O/r02_matdiv.txt 2>&1
cat $O/r02_matdiv.txt
timeout -k 10 600 python -m pytest tests/test_bench_launch.py -m gpu -x -q 2>&1 | tail -3
