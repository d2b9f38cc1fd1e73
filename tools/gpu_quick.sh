#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R/_r01 && echo "== r01" && python3 tools/per_tile_bench.py 300 2>&1 | grep -v amdgpu | tail -4 && python3 tools/progressive_bench.py 2>&1 | grep -v amdgpu | head -1
cd $R && echo "== r02" && python3 tools/per_tile_bench.py 300 2>&1 | grep -v amdgpu | tail -4 && python3 tools/progressive_bench.py 2>&1 | grep -v amdgpu | head -1
