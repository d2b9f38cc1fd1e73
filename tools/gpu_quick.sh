#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python3 bench.py --steps 5 2>$O/r02_f.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['frames_in_flight'], d['extra']['steps_mode'][:20], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['cpu_baseline']['value'])"
python3 bench.py --steps 5 --sync-steps --no-cpu-baseline 2>>$O/r02_f.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['frames_in_flight'], d['extra']['steps_mode'][:20])"
python3 bench.py --workload cfg5 --steps 2 --no-cpu-baseline 2>>$O/r02_f.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['frames_in_flight'], d['extra']['steps_mode'][:20])"
timeout -k 10 600 python -m pytest tests/test_bench_launch.py -m gpu -x -q 2>&1 | tail -2
