#!/bin/bash
# run quick_bench with every variant library present (GPU box)
for lib in yuki_amd/libyuki_hip_*.so; do
  n=$(basename $lib .so); n=${n#libyuki_hip_}
  echo "== $n"
  YK_LIB_PATH=$PWD/$lib python tools/quick_bench.py "$@" 2>&1 | tail -2 | head -1
done
