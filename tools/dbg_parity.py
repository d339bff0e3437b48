#!/usr/bin/env python3
"""Debug aid (GPU box): the bench's rig at a reduced size, parity of EVERY timed batch against the oracle, engine counters.
usage: dbg_parity.py [keys] [shards] [steps] [dispatch one|threads] [router placed|plain]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
import gubernator_amd as ga  # noqa: E402
import streams  # noqa: E402
import support  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 12
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 128
DISPATCH = sys.argv[4] if len(sys.argv) > 4 else "one"
ROUTER = sys.argv[5] if len(sys.argv) > 5 else "placed"

ctx = bench.Ctx()
ctx.world, ctx.rank, ctx.local_rank, ctx.dev = 1, 0, 0, torch.device("cuda", 0)
ctx.K, ctx.B = K, 65536
ctx.dispatch, ctx.streams, ctx.router = DISPATCH, 3, ROUTER
ctx.barrier = lambda: None
ctx.max_over_ranks = lambda v: v
ctx.table = streams.key_table(K)
ctx.my_ids = np.arange(K, dtype=np.int64)
NOW0 = streams.NOW0
rig = bench.Rig(ctx, "token", "zipf", S)
resident = rig.populate(NOW0)
print("resident", resident, "of", K, [e.stats()["cache_size"] for e in rig.engines], flush=True)
print("after populate:", [{k: e.stats()[k] for k in ("retries", "compactions", "unexpired_evictions", "tags_used")} for e in rig.engines][:3], flush=True)
rig.warmup, rig.steps, rig.profile_steps, rig.latency_steps = 8, STEPS, 0, 0
rig.build_stream(8 + STEPS, NOW0, 1234)
rig.keep_results(range(8, 8 + STEPS))
rig.run(0, 8)
rig.run(8, 8 + STEPS, timed=True)
for j, e in enumerate(rig.engines):
    st = e.stats()
    print(j, {k: st[k] for k in ("cache_size", "retries", "compactions", "unexpired_evictions", "tags_used", "batches", "fused_batches")}, flush=True)
orc = support.Oracle(cache_size=4 * K, workers=32)
bench.oracle_populate(rig, orc, 32, NOW0)
bad = 0
for s in range(0, 8 + STEPS):
    want = orc.eval(rig.host_batch(s), threads=32)
    if s in rig.kept:
        got = rig.kept[s].host()
        for name in ("status", "limit", "remaining", "reset_time", "err"):
            d = np.nonzero(getattr(got, name) != getattr(want, name))[0]
            if len(d):
                bad += 1
                i = int(d[0])
                print(f"batch {s} shard {rig.seq[s][0]}: {name} differs at {len(d)}; first idx {i} key id {int(rig.h_ids[s][i])} got",
                      (int(got.status[i]), int(got.limit[i]), int(got.remaining[i]), int(got.reset_time[i]), int(got.err[i])), "want",
                      (int(want.status[i]), int(want.limit[i]), int(want.remaining[i]), int(want.reset_time[i]), int(want.err[i])), flush=True)
                break
print("bad batches:", bad, "of", STEPS)
