#!/usr/bin/env python3
"""profiles/rNN_gather_bench.json from the raw output of tools/micro/gather_bench (mode 0 = what traversal does: every
lane fetches its own 64-byte record with four global_load_dwordx4) and, when present, the rocprofv3 TCP counter pass of
the same binary (tools/profile_gather.sh).  usage: gather_json.py raw.txt [pmc_dir] > out.json"""
import csv
import glob
import json
import os
import re
import sys

rows = {}
for line in open(sys.argv[1]):
    m = re.match(r"table (\d+) KB .* mode (\d): ([\d.]+) ms\s+([\d.]+) G records/s\s+([\d.]+) TB/s", line)
    if m and m.group(2) == "0":
        kb = int(m.group(1))
        rows[kb] = dict(table_bytes=kb * 1024, ms=float(m.group(3)), G_records_per_s=float(m.group(4)), GBps=float(m.group(5)) * 1e3)
if len(sys.argv) > 2:
    for d in sorted(glob.glob(os.path.join(sys.argv[2], "kb*"))):
        kb = int(os.path.basename(d)[2:])
        per = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k<0>" not in r["Kernel_Name"] and "ILi0E" not in r["Kernel_Name"]:
                    continue
                did = r["Dispatch_Id"]
                e = per.setdefault(did, dict(ns=int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
                e[r["Counter_Name"]] = float(r["Counter_Value"])
        if per and kb in rows:
            last = per[sorted(per, key=int)[-1]]  # the second (warm) repetition
            s = last["ns"] * 1e-9
            rows[kb]["pmc"] = {k: v for k, v in last.items() if k != "ns"}
            rows[kb]["pmc_ms"] = s * 1e3
            if "TCP_TOTAL_ACCESSES_sum" in last:
                rows[kb]["tcp_accesses_per_s"] = last["TCP_TOTAL_ACCESSES_sum"] / s
                rows[kb]["tcp_accesses_per_record"] = last["TCP_TOTAL_ACCESSES_sum"] / (256 * 6 * 256 * 2000)
cal = None
if len(sys.argv) > 2:
    for f in glob.glob(os.path.join(sys.argv[2], "fetch_kb*", "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if "k<0>" in r["Kernel_Name"] or "ILi0E" in r["Kernel_Name"]:
                per.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
        if per:
            last = per[sorted(per, key=int)[-1]]
            n_rec = 256 * 6 * 256 * 2000
            cal = dict(table_kb=int(re.search(r"fetch_kb(\d+)", f).group(1)), records=n_rec, counters=last,
                       fetch_size_bytes_per_record=last.get("FETCH_SIZE", 0.0) * 1024.0 / n_rec, ea_rdreq_per_record=last.get("TCC_EA0_RDREQ_sum", 0.0) / n_rec,
                       note="random 64-byte records of a 1 GB table (beyond L2 and Infinity Cache): FETCH_SIZE (KiB as reported) per record tells what one 64-byte gather costs in counted bytes")
out = dict(what="per-lane gathers of random 64-byte records (4 x global_load_dwordx4 per lane), 1536 blocks x 256 threads x 2000 dependent iterations, MI355X",
           source="tools/micro/gather_bench.hip mode 0; raw output beside this file", tables=[rows[k] for k in sorted(rows)], fetch_size_calibration=cal)
json.dump(out, sys.stdout, indent=1)
