#!/bin/bash
# randomised parity campaign on the GPU box: bash tools/gpu_fuzz.sh <first seed> <scenes> <stage scenes>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 ${4:-900} python3 tools/parity_fuzz.py $1 $2 > $O/r03_parity_fuzz_$1.txt 2>&1
tail -2 $O/r03_parity_fuzz_$1.txt
timeout -k 10 300 python3 tools/stage_fuzz.py $1 $3 > $O/r03_stage_fuzz_$1.txt 2>&1
tail -2 $O/r03_stage_fuzz_$1.txt
