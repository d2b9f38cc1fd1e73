#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r02_pytest_gpu_6.log 2>&1 || { tail -40 $O/r02_pytest_gpu_6.log; exit 1; }
tail -2 $O/r02_pytest_gpu_6.log
YK_DEBUG_BOUNCES=1 python3 tools/quick_bench.py cfg3 64 1920 1080 134217728 2>&1 | tail -10 | head -9 | awk '{print $1,$2, "shade", $(NF-4), "ms"}' | tr '\n' ';'; echo
python3 tools/quick_bench.py cfg3 64 1920 1080 134217728 2>&1 | tail -2 | head -1
python3 tools/shard_bench.py 2>&1 | tail -4
python3 tools/per_tile_bench.py 300 2>&1 | grep -v amdgpu | tail -4
python3 tools/progressive_bench.py 2>&1 | grep -v amdgpu | head -3
