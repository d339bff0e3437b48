#!/usr/bin/env python3
"""Per-kernel durations and stream overlap from a rocprofv3 --kernel-trace CSV (last N dispatches)."""
import csv, sys, collections, statistics
path, last = sys.argv[1], int(sys.argv[2])
rows = [r for r in csv.DictReader(open(path)) if "guber::" in r["Kernel_Name"]]
rows = rows[-last:]
per = collections.defaultdict(list)
for r in rows:
    per[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in per.items():
    print(f"{k:22s} n={len(d):5d} avg={sum(d)/len(d)/1e3:7.2f}us p50={statistics.median(d)/1e3:7.2f} max={max(d)/1e3:7.2f}")
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print(f"span {(t1-t0)/1e3:.1f}us, sum of kernel time {busy/1e3:.1f}us, avg concurrency {busy/(t1-t0):.2f}, per-batch period {(t1-t0)/1e3/(len(rows)/2):.2f}us")
