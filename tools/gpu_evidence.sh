#!/bin/bash
# The round's evidence set on one GPU box: GPU suite, bench lines of every single-GPU workload, cfg5's profile set.
# (cfg3's and cfg2's sets: tools/profile_round.sh r03_a cfg3 / r03_c cfg2, run separately.)  bash tools/gpu_evidence.sh <tag, e.g. r03>
T=${1:-r03}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/${T}_pytest_gpu_final.log 2>&1 || { tail -40 $O/${T}_pytest_gpu_final.log; exit 1; }
tail -2 $O/${T}_pytest_gpu_final.log
python3 bench.py --steps 5 > $O/${T}_a_bench.json 2> $O/${T}_a_bench.log
python3 bench.py --steps 5 --sync-steps --no-cpu-baseline > $O/${T}_a_bench_sync.json 2>> $O/${T}_a_bench.log
python3 bench.py --workload cfg2 --steps 5 > $O/${T}_c_bench_cfg2.json 2> $O/${T}_c_bench_cfg2.log
python3 bench.py --workload cfg1 --steps 20 > $O/${T}_c_bench_cfg1.json 2> $O/${T}_c_bench_cfg1.log
python3 bench.py --rccl-single --steps 5 --no-cpu-baseline > $O/${T}_d_rccl_single_abi.json 2> $O/${T}_d_rccl_single_abi.log
python3 bench.py --gpus 2 --steps 2 > $O/${T}_d_gpus2.json 2> $O/${T}_d_gpus2.log; echo "gpus2 rc=$?" >> $O/${T}_d_gpus2.log
python3 tools/write_scene_files.py cfg3 /tmp/yk_cfg3_files > /tmp/yk_cfg3_path.txt 2> $O/${T}_e_scene_file.log
python3 bench.py --scene-file $(cat /tmp/yk_cfg3_path.txt) --steps 3 --no-cpu-baseline > $O/${T}_e_bench_scene_file.json 2>> $O/${T}_e_scene_file.log
[ -n "$SKIP_CFG5_PROFILE" ] || bash tools/profile_round.sh ${T}_b cfg5
python3 bench.py --workload cfg5 --steps 3 > $O/${T}_b_bench.json 2> $O/${T}_b_bench.log
ls $O | grep "${T}_[a-e]"
