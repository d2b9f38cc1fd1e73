"""Device timeline of per-tile renders from a rocprofv3 kernel trace (csv): per tile — from one k_raygen to the next — the
span, the time some kernel was running, the idle gaps between dependent launches and the number of launches; then one tile's
launches in order.  What a hipGraph of the bounce loop could remove is bounded by the gaps.
usage: tile_timeline.py <kernel_trace.csv>"""
import csv, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]) for r in rows)
starts = [k for k, e in enumerate(ev) if "k_raygen" in e[2]]
tiles = []
for a, b in zip(starts[1:-1], starts[2:]):  # skip the warm-up tile
    seg = ev[a:b]
    busy_end, busy, gaps = seg[0][0], 0, 0
    for s, e, n in seg:
        if s > busy_end:
            gaps += s - busy_end
            busy_end = s
        if e > busy_end:
            busy += e - busy_end
            busy_end = e
    tiles.append((ev[b][0] - seg[0][0], busy, gaps, len(seg), seg))
n = len(tiles)
span = sum(t[0] for t in tiles) / n
busy = sum(t[1] for t in tiles) / n
gaps = sum(t[2] for t in tiles) / n
print(f"{n} tiles: {span / 1e3:.1f} us from raygen to raygen; a kernel running {busy / 1e3:.1f} us ({100 * busy / span:.0f} %), gaps inside the tile's "
      f"launch chain {gaps / 1e3:.1f} us ({100 * gaps / span:.0f} %), {(span - busy - gaps) / 1e3:.1f} us between the tile's last kernel and the next tile's raygen "
      f"(result copy, host turn-around); {sum(t[3] for t in tiles) / n:.0f} launches per tile")
per = defaultdict(lambda: [0, 0])
for t in tiles:
    for s, e, nme in t[4]:
        per[nme][0] += e - s
        per[nme][1] += 1
for nme, (d, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:8]:
    print(f"  {nme:46s} {d / n / 1e3:8.1f} us per tile in {c / n:4.1f} launches (avg {d / c / 1e3:6.1f} us)")
mid = tiles[n // 2][4]
print("one tile:")
prev = mid[0][0]
for s, e, nme in mid:
    print(f"  +{max(0, s - prev) / 1e3:6.1f} us gap, {(e - s) / 1e3:7.1f} us  {nme}")
    prev = max(prev, e)
