#!/bin/bash
# device timeline of the literal per-tile drop-in (tools/tile_loop.py under rocprofv3 --kernel-trace) -> gpurun_out/r02_tile_timeline.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python3 tools/tile_loop.py 60 > $O/r02_tile_timeline.txt 2>&1   # without the profiler
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tile_trace -o run -- python3 $R/tools/tile_loop.py 60 >> $O/r02_tile_timeline.txt 2>$O/tile_trace.log
f=$(find $O/tile_trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/tile_timeline.py $f >> $O/r02_tile_timeline.txt
rm -rf $O/tile_trace
cat $O/r02_tile_timeline.txt
