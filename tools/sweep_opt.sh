#!/bin/bash
# opt_bench (frame + 1/8 share) with the default library and every variant library present (GPU box)
echo "== default"; python tools/opt_bench.py shade_reorder 1 | tail -1
for lib in yuki_amd/libyuki_hip_*.so; do
  n=$(basename $lib .so); n=${n#libyuki_hip_}
  echo "== $n"
  YK_LIB_PATH=$PWD/$lib python tools/opt_bench.py shade_reorder 1 | tail -1
done
