#!/bin/bash
# ThreadSanitizer run of the tile combiner's host logic (yk_combiner.cpp + the stand-ins of tests/cpp/combiner_test.cpp).  CPU only.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${TMPDIR:-/tmp}/yk_tsan_combiner
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
hipcc -x hip --cuda-host-only -O1 -g -fsanitize=thread -std=c++17 -I include tests/cpp/combiner_test.cpp yuki_amd/csrc/yk_combiner.cpp -o $OUT/combiner_test_tsan -lpthread 2>/dev/null
LD_LIBRARY_PATH=/opt/rocm/lib TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1" $OUT/combiner_test_tsan
