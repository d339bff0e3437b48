#!/usr/bin/env python3
"""guber_global_sync volume timing on one GPU: R logical ranks, K GLOBAL keys hit on every rank between syncs.
usage: bench_global_native.py [ranks] [keys] [rounds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gubernator_amd as ga
from gubernator_amd import global_native as gn
import streams
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ring = ga.Ring([f"gpu{i}" for i in range(R)])
cl = gn.Comm.local([ga.Engine(cache_size=2 * K, max_batch=65536, max_key_bytes=64, flags=ga.FLAG_GLOBAL) for _ in range(R)], ring)
tab = streams.key_table(K)
now = streams.NOW0
ms = []
for rnd in range(rounds):
    for r in range(R):
        for lo in range(0, K, 65536):
            kb, ko = streams.keys_for_ids(tab, np.arange(lo, min(K, lo + 65536)))
            cl.ranks[r].evaluate((kb, ko), 1, 1000, 600_000, now)
    t0 = time.perf_counter()
    st = cl.sync(now)
    ms.append((time.perf_counter() - t0) * 1e3)
    now += 10
print({"ranks": R, "keys": K, "rows_per_sync": {k: sum(s_[k] for s_ in st) for k in ("hits_sent", "hits_applied", "broadcast", "installed")},
       "bytes_moved": cl.last["bytes_moved"], "ms": [round(x, 3) for x in ms]})
