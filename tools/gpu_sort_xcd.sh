#!/bin/bash
# XCD-affine claims on Morton-sorted queues (tools/micro/ray_sort_experiment.h; build: tools/build_variant.sh sortxcd -DYK_EXPERIMENT_SORT -DYK_EXPERIMENT_XCD)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_sort; mkdir -p $O; cd $R
export YK_LIB_PATH=$R/yuki_amd/libyuki_hip_sortxcd.so YK_DEBUG_BOUNCES=1
for cfg in "0 9 2" "4 9 2" "4 9 0" "4 5 2"; do set -- $cfg
  YK_SORT_BOUNCES=$1 YK_SORT_BITS=$2 YK_SORT_MODE=$3 YK_SORT_SHADOW=1 python3 tools/quick_bench.py cfg3 64 1920 1080 134217728 > $O/xcd_k$1_b$2_m$3.txt 2>&1
  echo "== XCD-affine claims, sort bounces=$1 bits=$2 mode=$3"; grep -E "^bounce [0-4]|^mean" $O/xcd_k$1_b$2_m$3.txt | tail -6
done
