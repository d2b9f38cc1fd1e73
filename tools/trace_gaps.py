"""Summarise a rocprofv3 kernel trace CSV: per kernel name total/avg, and the idle gaps of the
device timeline (time with no kernel running).  usage: trace_gaps.py <kernel_trace.csv> [t_from_frac]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo = t0 + (t1 - t0) * float(sys.argv[2]) if len(sys.argv) > 2 else t0
ev = [e for e in ev if e[0] >= lo]
busy_end = ev[0][0]
idle = 0
gaps = []
for s, e, n in ev:
    if s > busy_end:
        idle += s - busy_end
        gaps.append((s - busy_end, n))
    busy_end = max(busy_end, e)
span = busy_end - ev[0][0]
print(f"span {span/1e6:.2f} ms, idle {idle/1e6:.2f} ms ({100*idle/span:.1f}%), {len(ev)} launches")
gaps.sort(reverse=True)
from collections import defaultdict
by = defaultdict(lambda: [0, 0])
for g, n in gaps:
    by[n][0] += g; by[n][1] += 1
for n, (g, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  idle before {n:60s} {g/1e6:8.3f} ms in {c} gaps (avg {g/c/1e3:.1f} us)")
tot = defaultdict(lambda: [0, 0])
for s, e, n in ev:
    tot[n][0] += e - s; tot[n][1] += 1
for n, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  {n:60s} {d/1e6:8.3f} ms in {c} launches (avg {d/c/1e3:.1f} us)")
