#!/bin/bash
# ThreadSanitizer run of the parallel host BVH builder (yk_host.cpp) on 200 k random boxes, every split method.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${TMPDIR:-/tmp}/yk_tsan
mkdir -p $OUT
cd $ROOT/yuki_amd/csrc
hipcc -x hip --cuda-host-only -std=c++17 -O1 -g -fsanitize=thread -I. -I../../include yk_host.cpp $ROOT/tools/asan/tsan_bvh.cpp -o $OUT/tsan_bvh -pthread
YK_BVH_THREADS=${1:-6} $OUT/tsan_bvh
