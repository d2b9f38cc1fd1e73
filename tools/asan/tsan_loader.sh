#!/bin/bash
# ThreadSanitizer run of the pbrt loader's parallel PLY stage (yk_loaders.cpp: plymesh files are read by a thread pool after the
# parse).  CPU only: the host files are built with the host half of hipcc, the harness loads a city scene written as
# pbrt + one PLY per mesh (tests/scene_files.py) a few times.   usage: tools/asan/tsan_loader.sh [scene=city-small]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${TMPDIR:-/tmp}/yk_tsan_loader
rm -rf $OUT; mkdir -p $OUT
cd $ROOT/yuki_amd/csrc
hipcc -x hip --cuda-host-only -O1 -g -fsanitize=thread -std=c++17 -fPIC -ffp-contract=off -shared -o $OUT/libyk_host_tsan.so yk_loaders.cpp yk_image.cpp yk_image_formats.cpp yk_host.cpp 2>/dev/null
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -fsanitize=thread -std=c++17 $ROOT/tools/asan/harness.cpp -o $OUT/harness -L$OUT -lyk_host_tsan -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$OUT
cd $ROOT && python - "$OUT/files" "${1:-city-small}" <<'P'
import sys
sys.path.insert(0, "tests")
import scene_files as sf
from yuki_amd import scenes
print(sf.write_scene_as_pbrt(sys.argv[1], scenes.by_name(sys.argv[2]), res=(320, 180)))
P
cd $OUT && LD_LIBRARY_PATH=/opt/rocm/lib TSAN_OPTIONS=halt_on_error=0 ./harness files/scene.pbrt files/scene.pbrt files/scene.pbrt
