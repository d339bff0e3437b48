#!/usr/bin/env python3
"""What the GPU does during a run of generations through the front: from a rocprofv3 --kernel-trace CSV, the longest stretch of guber
kernels without a gap of more than 200 us (the one call of many generations), per kernel: launches, average duration, share of the
stretch; how many kernels run at a time; and per queue how much of the stretch it is busy.
    python tools/front_timeline.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "guber::" in r["Kernel_Name"]]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("guber::", ""), r.get("Queue_Id", "?")) for r in rows)
# split into stretches at gaps > 200 us
stretches, cur, end = [], [], None
for e in ev:
    if end is not None and e[0] - end > 200_000:
        stretches.append(cur); cur = []
    cur.append(e); end = max(end or 0, e[1])
stretches.append(cur)
best = max(stretches, key=len)
t0, t1 = best[0][0], max(e[1] for e in best)
span = t1 - t0
per, perq = collections.defaultdict(list), collections.defaultdict(int)
for s, e, k, q in best:
    per[k].append(e - s); perq[q] += e - s
print(f"stretch: {len(best)} launches in {span / 1e3:.1f} us")
for k, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:20s} n={len(d):5d} avg={sum(d) / len(d) / 1e3:7.2f} us  max={max(d) / 1e3:7.2f}  busy={sum(d) / span:5.2f} of the stretch")
# concurrency histogram
pts = sorted([(s, 1) for s, e, k, q in best] + [(e, -1) for s, e, k, q in best])
hist, lvl, last = collections.defaultdict(int), 0, t0
for t, d in pts:
    hist[lvl] += t - last; last = t; lvl += d
print("  kernels running at a time:", {k: round(v / span, 3) for k, v in sorted(hist.items())})
print("  per queue busy:", {q: round(v / span, 2) for q, v in sorted(perq.items())})
