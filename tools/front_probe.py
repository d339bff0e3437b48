#!/usr/bin/env python3
"""the device-resident front alone (guber_front_*): generations one at a time (per-kernel HIP-event times with nothing else in flight),
then a run of generations in one call (wall clock).  A measurement aid for scripts/gpu_r06_*.sh, often run under rocprofv3 --kernel-trace.
    python tools/front_probe.py [keys] [gen_batches] [generations] [shards] [streams]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import bench
import streams

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
GB = int(sys.argv[2]) if len(sys.argv) > 2 else 8
NG = int(sys.argv[3]) if len(sys.argv) > 3 else 32
S = int(sys.argv[4]) if len(sys.argv) > 4 else 12
NS = int(sys.argv[5]) if len(sys.argv) > 5 else 3
ctx = bench.Ctx()
ctx.world, ctx.rank, ctx.local_rank, ctx.dev = 1, 0, 0, torch.device("cuda", 0)
torch.cuda.set_device(0)
ctx.K, ctx.B, ctx.dispatch, ctx.streams, ctx.router = K, 65536, "one", NS, "placed"
ctx.barrier = lambda: None
ctx.max_over_ranks = lambda v: v
ctx.table = streams.key_table(K)
ctx.my_ids = np.arange(K, dtype=np.int64)
rig = bench.RoutedRig(ctx, "token", os.environ.get("PROBE_DIST", "zipf"), S, GB)
rig.populate(streams.NOW0)
idle = 8
rig.warmup, rig.steps, rig.profile_steps, rig.latency_steps = 4 * GB, NG * GB, idle * GB, 0
rig.build_stream((4 + NG + idle) * GB, streams.NOW0, 1234)
rig.run(0, 4 * GB)
t, _ = rig.run(4 * GB, (4 + NG) * GB, timed=True)
print(f"{NG} generations of {GB} x 65536 in one call: {t * 1e3:.3f} ms = {NG * GB * 65536 / t / 1e9:.3f} G decisions/s; front {rig.front.stats()}")
for e in rig.engines:
    e.profile(True)
    e.profile_read()
for g in range(4 + NG, 4 + NG + idle):
    rig.run(g * GB, (g + 1) * GB)
tot = {}
for e in rig.engines:
    for k, (cnt, ms) in e.profile_read().items():
        if cnt:
            a = tot.setdefault(k, [0, 0.0])
            a[0] += cnt
            a[1] += ms
print("one generation at a time, per launch (us):", {k: round(v[1] / v[0] * 1e3, 1) for k, v in tot.items()}, "launches", {k: v[0] for k, v in tot.items()})
print("generation latencies (us):", [round(x, 1) for x in rig.front.latencies()])
rig.close()
