cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gapt -o one -- python3 tools/per_tile_bench.py 60 > gpurun_out/gapt_run.log 2>&1
f=$(find gpurun_out/gapt -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f 0.0 > gpurun_out/gapt_summary.txt
python - "$f" <<'PY' >> gpurun_out/gapt_summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]) for r in rows)
# one tile of the single-thread phase: find k_raygen launches and print the span between two consecutive ones in the middle
rg = [i for i, e in enumerate(ev) if "k_raygen" in e[2]]
a, b = rg[20], rg[21]
print("one tile: %.3f ms between raygen launches, %d launches, kernel time %.3f ms" % ((ev[b][0] - ev[a][0]) / 1e6, b - a, sum(e[1] - e[0] for e in ev[a:b]) / 1e6))
prev = ev[a][0]
for s, e, n in ev[a:b]:
    print("  +%7.1f us gap, %7.1f us  %s" % ((s - prev) / 1e3, (e - s) / 1e3, n))
    prev = e
PY
rm -rf gpurun_out/gapt
