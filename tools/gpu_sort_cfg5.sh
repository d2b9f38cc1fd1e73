#!/bin/bash
# The ray-order experiment on the scene that does NOT fit the Infinity Cache: cfg5's 10,240,012 triangles (1.72 GB of nodes and
# triangles), 1920x1080 x 64 spp, Path 8, one batch (tools/micro/ray_sort_experiment.h).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_sort; mkdir -p $O; cd $R
export YK_DEBUG_BOUNCES=1
run() { n=$1; lib=$2; shift 2
  env YK_LIB_PATH=$R/yuki_amd/libyuki_hip_$lib.so "$@" python3 tools/quick_bench.py cfg5 64 1920 1080 134217728 > $O/cfg5_$n.txt 2>&1
  echo "== cfg5 $n"; grep -E "^bounce [0-4]|sort after shade [0-2]|^wall|^mean" $O/cfg5_$n.txt | tail -10; }
run base sort YK_SORT_BOUNCES=0
run k4_b9_m0_sh sort YK_SORT_BOUNCES=4 YK_SORT_BITS=9 YK_SORT_MODE=0 YK_SORT_SHADOW=1
run k4_b9_m2_sh sort YK_SORT_BOUNCES=4 YK_SORT_BITS=9 YK_SORT_MODE=2 YK_SORT_SHADOW=1
run xcd_k4_b9_m2_sh sortxcd YK_SORT_BOUNCES=4 YK_SORT_BITS=9 YK_SORT_MODE=2 YK_SORT_SHADOW=1
run xcd_k0 sortxcd YK_SORT_BOUNCES=0
