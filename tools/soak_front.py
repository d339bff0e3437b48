#!/usr/bin/env python3
"""Randomised soak of guber_front_* against the oracle — on the CPU build of the engine (tests/hostsim/libenginesim.so: host code + kernels compiled
for the host) or, without GUBER_HIP_LIB, on the GPU through the product library: engines x streams x max_batch x generation size x key population x key form x which columns are present x binding caches.
Test infrastructure.
    GUBER_HIP_LIB=tests/hostsim/libenginesim.so python tools/soak_front.py [seconds] [seed]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import gubernator_amd as ga
import streams
import support

ON_GPU = "enginesim" not in ga.LIB_PATH                             # (the product library: real streams and events — what the CPU build cannot show)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sys.path.insert(0, os.path.join(ROOT, "tests"))
if ON_GPU:
    import torch
    from test_gpu_front import dev_gen as _dev_gen_gpu
    _dev = torch.device("cuda", 0)

    def dev_gen(hb, full):
        b, res, t, r = _dev_gen_gpu(torch, _dev, hb, full)
        return b, res, t, r
else:
    os.environ.setdefault("GUBER_HIP_LIB", ga.LIB_PATH)
    from enginesim_cases import dev_gen   # noqa: E402

t_end, it = time.time() + budget, 0
while time.time() < t_end:
    seed = seed0 * 100_000 + it
    rng = np.random.default_rng(seed)
    ne = int(rng.integers(1, 7)); ns = int(rng.integers(1, min(ne, 3) + 1)); mb = int(rng.choice([1024, 2048, 4096])); G = int(rng.choice([1500, 4096, 8192]))
    K = int(rng.choice([300, 5000, 40000])); bind = bool(rng.random() < 0.3); fixed = bool(rng.random() < 0.6); full = bool(rng.random() < 0.4)
    cs = max(ne * 64, K // 2) if bind else 1 << 17
    cfg = dict(seed=seed, engines=ne, streams=ns, max_batch=mb, G=G, keys=K, binding=bind, fixed_width=fixed, full_columns=full, cache=cs)
    place = ga.Placement(ne) if ne > 1 else None
    if place is not None and not bind:                               # (a binding cache is compared with the oracle's workers: the placement stays the worker rule)
        tab0 = streams.key_table(K)
        place.observe_keys(*streams.keys_for_ids(tab0, streams.ZipfSampler(K, seed=seed).draw(1 << 13)))
        place.rebalance(0.125, True)
    engs = []
    for j in range(ne):
        sj = j * ns // ne
        first = next((q for q in range(j) if q * ns // ne == sj), None)
        engs.append(ga.Engine(cache_size=(cs // ne) if bind else cs, max_batch=mb, stream=None if first is None else engs[first].stream_handle()))
    fr = ga.Front(engs, place, max_n=G, depth=int(rng.integers(3, 7)))
    orc = support.Oracle(cache_size=cs if bind else 1 << 20, workers=ne if bind else 1)
    tab = streams.key_table(K)
    now = streams.NOW0
    try:
        gens = []
        for g in range(int(rng.integers(2, 7))):
            n = int(rng.choice([0, 1, 3, G, G, int(rng.integers(1, G + 1))]))
            ids = rng.integers(0, K, n) if rng.random() < 0.5 else (rng.zipf(1.3, n) % K)
            if fixed:
                hb = streams.bench_batch(tab, ids, now, algorithm=0, limit=int(rng.choice([3, 30])), duration=int(rng.choice([900, 60000])))
            else:
                hb = support.HostBatch([f"k{int(i)}" + "y" * int(i % 9) for i in ids], 1, 30, 60000, now)
            if n:
                hb.algorithm[:] = (ids + g) % 2
                if rng.random() < 0.3:
                    hb.behavior[:] = np.where(rng.random(n) < 0.05, 8, 0).astype(np.uint32)     # some RESET_REMAINING
            if full and n:
                hb = support.HostBatch((hb.key_bytes, hb.key_off), hb.hits, hb.limit, hb.duration, now, burst=np.where(ids % 4 == 0, 5, 0), created_at=now - (ids % 3),
                                       algorithm=hb.algorithm, behavior=hb.behavior, is_owner=(ids % 5 != 0).astype(np.uint8))
            gens.append(hb)
            now += int(rng.choice([0, 1, 400, 1200]))
        parts = [dev_gen(hb, full) for hb in gens]
        if ON_GPU:
            torch.cuda.synchronize(_dev)
        N = len(gens)
        # in one call or one by one
        if rng.random() < 0.5:
            assert fr.eval_dev((ga.GuberBatch * N)(*[x[0] for x in parts]), (ga.GuberResult * N)(*[x[1] for x in parts]), N) == N
        else:
            for x in parts:
                assert fr.eval_dev((ga.GuberBatch * 1)(x[0]), (ga.GuberResult * 1)(x[1]), 1) == 1
        fr.synchronize()
        if ON_GPU:
            parts = [(x[0], x[1], x[2], {kk: vv.cpu().numpy()[:gens[q].n] for kk, vv in x[3].items()}) for q, x in enumerate(parts)]
        for k, hb in enumerate(gens):
            want = orc.eval(hb)
            if hb.n:
                got = ga.HostResult(hb.n)
                for name in ("status", "limit", "remaining", "reset_time", "err"):
                    getattr(got, name)[:hb.n] = parts[k][3][name]
                support.assert_results_equal(got, want, f"generation {k}")
        assert sum(e.size() for e in engs) == orc.size(), ([e.size() for e in engs], orc.size())
    except Exception:
        print("FAILED configuration:", cfg, flush=True)
        raise
    finally:
        fr.close()
        for e in engs:
            e.close()
        if place is not None:
            place.close()
        orc.close()
    it += 1
print(f"soak_front: {it} configurations, no difference")
