cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap -o one -- python3 tools/overlap_frames.py 8 16 one > gpurun_out/gap_run.log 2>&1
f=$(find gpurun_out/gap -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f 0.85 > gpurun_out/gap_summary.txt
rm -rf gpurun_out/gap
