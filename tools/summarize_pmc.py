#!/usr/bin/env python3
"""Average every counter of gpurun_out/pmc2_*/pmc_counter_collection.csv per kernel over the last N dispatches."""
import csv, glob, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
out = collections.defaultdict(dict)
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc2_*", "pmc_counter_collection.csv"))):
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0]
        if "guber::" in name:
            per[(name.replace("guber::", ""), row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in per.items():
        v = v[-n:]
        out[k][c] = sum(v) / len(v)
for k in sorted(out):
    print(k)
    for c in sorted(out[k]):
        print(f"    {c:32s} {out[k][c]:16.1f}")
