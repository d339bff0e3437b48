#!/usr/bin/env python3
"""BASELINE configs[0] (benchmark_test.go:63-84 shape): TOKEN_BUCKET, ONE request per call through the C ABI
(guber_eval_batch, host pointers): (i) 1 000 pre-generated keys cycled, (ii) a fresh key per op.  Reports the per-call
latency, i.e. the boundary + launch overhead a non-batched caller pays."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import gubernator_amd as ga
from gubernator_amd.abi import HostBatch

e = ga.Engine(cache_size=1 << 20, max_batch=1024)
now = 1_700_000_000_000
for label, keyfn in (("1000 keys cycled", lambda i: b"bench_%04d" % (i % 1000)), ("fresh key per op", lambda i: b"fresh_%09d" % i)):
    batches = [HostBatch([keyfn(i)], 1, 10, 5000, now + i) for i in range(3000)]
    for b in batches[:200]:
        e.eval(b)
    lat = []
    for b in batches[200:]:
        t0 = time.perf_counter(); e.eval(b); lat.append((time.perf_counter() - t0) * 1e6)
    lat.sort()
    print(f"batch = 1, {label:18s}: p50 {lat[len(lat)//2]:7.1f} us  p99 {lat[int(len(lat)*0.99)]:7.1f} us  -> {1e6/ (sum(lat)/len(lat)):9.0f} decisions/s per caller thread")
