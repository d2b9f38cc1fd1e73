#!/bin/bash
# per-bounce breakdown of the full cfg3 frame and of its 1/8 share (YK_DEBUG_BOUNCES synchronises: diagnostics only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
YK_DEBUG_BOUNCES=1 python3 tools/quick_bench.py cfg3 64 1920 1080 134217728 > $O/r02_bounces_full.txt 2>&1
YK_DEBUG_BOUNCES=1 python3 tools/shard8_debug.py 2 > $O/r02_bounces_shard8.txt 2>&1
python3 tools/shard_bench.py > $O/r02_shard_bench.txt 2>&1
python3 tools/overlap_frames.py 8 16 2 > $O/r02_overlap_shard8.txt 2>&1
python3 tools/per_tile_bench.py > $O/r02_per_tile.txt 2>&1
python3 tools/progressive_bench.py > $O/r02_progressive.txt 2>&1
tail -5 $O/r02_shard_bench.txt; cat $O/r02_overlap_shard8.txt $O/r02_per_tile.txt $O/r02_progressive.txt
