#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests/test_bench_launch.py tests/test_multi_gpu.py tests/test_cxx_mirror.py -m gpu -x -q 2>&1 | tail -3
echo "rc=$?"
timeout -k 10 600 python -m pytest tests/test_multi_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
