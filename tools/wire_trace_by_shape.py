"""Kernel time per device wire decode BY SHAPE from a rocprofv3 kernel trace of tests/test_gpu_wire_dev.py's throughput test (the shape =
k_wire_scan's grid: one workgroup per payload).  usage: wire_trace_by_shape.py <kernel_trace.csv>"""
import csv, sys, collections, statistics
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_wire" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dec, cur = [], []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1]
    if name in ("k_wire_win_a", "k_wire_win_b") and cur and any(n in ("k_wire_scan", "k_wire_fill", "k_wire_kill") for n, _, _ in cur): dec.append(cur); cur = []
    if name == "k_wire_scan" and cur and any(n in ("k_wire_fill", "k_wire_kill") for n, _, _ in cur): dec.append(cur); cur = []
    cur.append((name, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))))
if cur: dec.append(cur)
by = collections.defaultdict(list)
for d in dec:
    nrpc = next((g // 64 for n, _, g in d if n == "k_wire_scan"), 0)
    by[nrpc].append(d)
items = {64: 64000, 640: 64000, 4000: 40000}
for nrpc, ds in sorted(by.items()):
    tot = [sum(t for _, t, _ in d) for d in ds]
    per = collections.defaultdict(list)
    for d in ds:
        for n, t, _ in d: per[n].append(t)
    med = statistics.median(tot)
    line = f"{nrpc} payloads: {len(ds)} decodes, kernel time per decode median {med / 1e3:.1f} us"
    if nrpc in items: line += f" = {items[nrpc] / med * 1e3:.0f} M items/s"
    print(line, {n: round(statistics.median(v) / 1e3, 1) for n, v in per.items()})
