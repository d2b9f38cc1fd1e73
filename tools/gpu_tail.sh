#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python3 tools/opt_bench.py tail_paths 0 16384 65536 262144 524288 1048576 2097152 > $O/r02_tail_sweep.txt 2>&1
cat $O/r02_tail_sweep.txt
for t in 0 16384 65536 262144; do echo "== tail_paths=$t"; YK_OPTS=tail_paths=$t python3 tools/per_tile_bench.py 300 2>&1 | grep -v amdgpu.ids; done > $O/r02_tail_per_tile.txt 2>&1
cat $O/r02_tail_per_tile.txt
for t in 0 65536 262144 1048576; do echo "== tail_paths=$t"; YK_TAIL_PATHS=$t python3 tools/progressive_bench.py 2>&1 | grep -v amdgpu.ids | head -3; done > $O/r02_tail_progressive.txt 2>&1
cat $O/r02_tail_progressive.txt
