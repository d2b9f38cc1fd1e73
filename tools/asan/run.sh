#!/bin/bash
# AddressSanitizer + leak check of the host-only parsers (PLY, pbrt-v3, PNG) on mutated inputs.
# CPU only (GPU ASan is not available on this pool): the three host files are built with the
# host half of hipcc, linked into a small harness and fed tests/test_loader_fuzz.py's corpus.
#   usage: tools/asan/run.sh [seed] [count]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${TMPDIR:-/tmp}/yk_asan
mkdir -p $OUT/corpus
cd $ROOT/yuki_amd/csrc
hipcc -x hip --cuda-host-only -O1 -g -fsanitize=address -shared-libasan -std=c++17 -fPIC -ffp-contract=off -shared -o $OUT/libyk_host_asan.so yk_loaders.cpp yk_image.cpp yk_image_formats.cpp yk_host.cpp 2>/dev/null
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -fsanitize=address -shared-libasan -std=c++17 $ROOT/tools/asan/harness.cpp -o $OUT/harness -L$OUT -lyk_host_asan -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$OUT
rm -f $OUT/corpus/fz*
cd $ROOT && python -c "
import sys; sys.path.insert(0, 'tests')
import test_loader_fuzz as t
t.write_corpus('$OUT/corpus', seed=${1:-1}, count=${2:-4000})"
RT=$(dirname $(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so))
cd $OUT && LD_LIBRARY_PATH=$RT:/opt/rocm/lib ASAN_OPTIONS=detect_leaks=1:allocator_may_return_null=1 ./harness corpus/fz*
