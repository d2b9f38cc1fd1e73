#!/bin/bash
# Ray-order experiment (tools/micro/ray_sort_experiment.h; build: tools/build_variant.sh sort -DYK_EXPERIMENT_SORT).
# Per-bounce kernel times of the cfg3 frame with the queues entering bounces 1..k sorted by (origin Morton, octant).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_sort; mkdir -p $O
cd $R
export YK_LIB_PATH=$R/yuki_amd/libyuki_hip_sort.so YK_DEBUG_BOUNCES=1
run() { # name, env...
  n=$1; shift
  env "$@" python3 tools/quick_bench.py cfg3 64 1920 1080 134217728 > $O/$n.txt 2>&1 || return 1
  tail -14 $O/$n.txt | grep -E "^bounce [0-4]|sort after|wall" | tail -12
  echo "== $n done"
}
run base YK_SORT_BOUNCES=0 &&
run k4_b9_m0 YK_SORT_BOUNCES=4 YK_SORT_BITS=9 YK_SORT_MODE=0 &&
run k4_b9_m0_sh YK_SORT_BOUNCES=4 YK_SORT_BITS=9 YK_SORT_MODE=0 YK_SORT_SHADOW=1 &&
run k4_b9_m1_sh YK_SORT_BOUNCES=4 YK_SORT_BITS=9 YK_SORT_MODE=1 YK_SORT_SHADOW=1 &&
run k4_b9_m2_sh YK_SORT_BOUNCES=4 YK_SORT_BITS=9 YK_SORT_MODE=2 YK_SORT_SHADOW=1 &&
run k4_b5_m0_sh YK_SORT_BOUNCES=4 YK_SORT_BITS=5 YK_SORT_MODE=0 YK_SORT_SHADOW=1 &&
run k4_b3_m0_sh YK_SORT_BOUNCES=4 YK_SORT_BITS=3 YK_SORT_MODE=0 YK_SORT_SHADOW=1 &&
run k4_m3_sh YK_SORT_BOUNCES=4 YK_SORT_MODE=3 YK_SORT_SHADOW=1
