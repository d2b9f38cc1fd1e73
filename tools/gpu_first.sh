#!/bin/bash
# first GPU call of round 2: counter list, gather ceiling at several table sizes, baseline tests + bench
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/r02_counters.txt 2>&1 || true
cd $R
( for kb in 64 1024 8192 32768 102400 262144 1048576 2097152; do ./tools/micro/gather_bench $kb 2000; done ) > $O/r02_gather_bench_raw.txt 2>&1
python -m pytest tests -m gpu -x -q > $O/r02_pytest_gpu_0.log 2>&1
python3 bench.py --steps 5 > $O/r02_0_bench.json 2> $O/r02_0_bench.log
