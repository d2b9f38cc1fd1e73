#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r02_pytest_gpu_final.log 2>&1 || { tail -40 $O/r02_pytest_gpu_final.log; exit 1; }
tail -2 $O/r02_pytest_gpu_final.log
bash tools/profile_round.sh r02_a cfg3
cp $O/r02_a_pmc_cfg3.json profiles/
python3 bench.py --steps 5 > $O/r02_a_bench.json 2> $O/r02_a_bench.log
python3 bench.py --steps 5 --no-cpu-baseline --two-in-flight > $O/r02_a_two_in_flight.json 2>> $O/r02_a_bench.log
bash tools/profile_round.sh r02_b cfg5
cp $O/r02_b_pmc_cfg5.json profiles/
python3 bench.py --workload cfg5 --steps 3 > $O/r02_b_bench.json 2> $O/r02_b_bench.log
python3 bench.py --workload cfg2 --steps 5 > $O/r02_c_bench_cfg2.json 2> $O/r02_c_bench_cfg2.log
python3 bench.py --workload cfg1 --steps 20 > $O/r02_c_bench_cfg1.json 2> $O/r02_c_bench_cfg1.log
python3 bench.py --rccl-single --steps 5 --no-cpu-baseline > $O/r02_d_rccl_single_abi.json 2> $O/r02_d_rccl_single_abi.log
python3 bench.py --rccl-single --gather torch --steps 5 --no-cpu-baseline > $O/r02_d_rccl_single_torch.json 2> $O/r02_d_rccl_single_torch.log
python3 bench.py --gpus 2 --steps 2 > $O/r02_d_gpus2.json 2> $O/r02_d_gpus2.log; echo "gpus2 rc=$?" >> $O/r02_d_gpus2.log
ls $O | grep "r02_[a-d]"
