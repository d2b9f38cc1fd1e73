#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r02_pytest_gpu_2.log 2>&1 || { tail -40 $O/r02_pytest_gpu_2.log; exit 1; }
tail -2 $O/r02_pytest_gpu_2.log
python3 bench.py --steps 5 > $O/r02_a_bench.json 2> $O/r02_a_bench.log
tail -c 1500 $O/r02_a_bench.json
python3 bench.py --workload cfg5 --steps 3 > $O/r02_b_bench_pre.json 2> $O/r02_b_bench_pre.log
bash tools/profile_round.sh r02_b cfg5
cp $O/r02_b_pmc_cfg5.json profiles/
python3 bench.py --workload cfg5 --steps 3 > $O/r02_b_bench.json 2> $O/r02_b_bench.log
ls $O | grep r02_b
