R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_sort; mkdir -p $O; cd $R
export YK_LIB_PATH=$R/yuki_amd/libyuki_hip_sort.so YK_DEBUG_BOUNCES=1 YK_PACKET_BOUNCES=2
for cfg in "0 9 0" "1 9 0" "1 9 1" "1 5 1" "1 3 0"; do set -- $cfg
  YK_SORT_BOUNCES=$1 YK_SORT_BITS=$2 YK_SORT_MODE=$3 python3 tools/quick_bench.py cfg3 64 1920 1080 134217728 > $O/pkt2_k$1_b$2_m$3.txt 2>&1
  echo "== packets on bounce 1, sort=$1 bits=$2 mode=$3"; grep -E "^bounce 1" $O/pkt2_k$1_b$2_m$3.txt | tail -1
done
