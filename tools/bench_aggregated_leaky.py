#!/usr/bin/env python3
"""Latency of one 65536-request LEAKY batch on a Zipf-1.1 key stream when the batch aggregates 66 RPC payloads whose
requests are stamped 0..3 ms apart (created_at per 1000-item slice) versus one common created_at.  Shows what the
created_at-harmless rule (guber_algo.h leaky_created_harmless) buys for hot leaky keys."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.cuda.init()
import gubernator_amd as ga
from gubernator_amd.abi import HostBatch
import streams

K, B = 1_000_000, 65536
e = ga.Engine(cache_size=K, max_batch=B)
kt = streams.key_table(K)
z = streams.ZipfSampler(K)
now = streams.NOW0
for mode in ("uniform created_at", "per-RPC created_at (+0..3 ms)", "per-RPC created_at, one slice 700 ms late"):
    lat = []
    for it in range(12):
        ids = z.draw(B)
        kb, ko = streams.keys_for_ids(kt, ids)
        created = np.full(B, now, np.int64)
        if mode != "uniform created_at":
            created = now - 2 + (np.arange(B) // 1000) % 4
        if mode.endswith("late"):
            created[5000:6000] += 700
        b = HostBatch((kb, ko), 1, 100, 60_000, now, created_at=created, algorithm=1)
        t0 = time.perf_counter(); e.eval(b); t1 = time.perf_counter()
        lat.append((t1 - t0) * 1e6)
        now += 1
    lat.sort()
    print(f"{mode:48s} host-staged batch: p50 {lat[len(lat)//2]:9.1f} us   min {lat[0]:9.1f} us")
