#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r02_pytest_gpu_1.log 2>&1 || { tail -40 $O/r02_pytest_gpu_1.log; exit 1; }
tail -3 $O/r02_pytest_gpu_1.log
bash tools/profile_gather.sh r02
bash tools/profile_round.sh r02_a cfg3
ls $O | grep r02
