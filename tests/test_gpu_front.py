"""guber_front_* on the GPU (include/guber_gpu.h "the front of a GPU's logical shards"): one stream of requests in ARRIVAL order, resident
in HBM -> routed on the device (XXH64 + the placement's rule: WorkerPool.getWorker, workers.go:153-155,180-184) -> the engines' fused
pipelines -> the answers in ARRIVAL order (gubernator.proto:51-54), against ONE oracle fed the same requests one by one in arrival order
(gubernator.go:203).  The same cases run on the CPU build of the engine in tests/test_enginesim_cpu.py."""
import ctypes as C

import numpy as np
import pytest

import gubernator_amd as ga
import streams
import support
from support import Oracle

pytestmark = pytest.mark.gpu


def dev_gen(torch, dev, hb, full):
    n = hb.n
    cols = dict(key_bytes=np.concatenate([hb.key_bytes, np.zeros(16, np.uint8)]), key_off=hb.key_off.view(np.int32), hits=hb.hits, limit=hb.limit,
                duration=hb.duration, algorithm=hb.algorithm, behavior=hb.behavior.view(np.int32),
                burst=hb.burst if full else None, created_at=hb.created_at if full else None, is_owner=hb.is_owner if full else None)   # (None: the column is absent)
    t = {k: (torch.from_numpy(np.ascontiguousarray(v)).to(dev) if v is not None else None) for k, v in cols.items()}
    p = {k: (v.data_ptr() if v is not None else None) for k, v in t.items()}
    r = dict(status=torch.full((max(n, 1),), 99, dtype=torch.uint8, device=dev), err=torch.full((max(n, 1),), 99, dtype=torch.uint8, device=dev),
             limit=torch.full((max(n, 1),), -7, dtype=torch.int64, device=dev), remaining=torch.full((max(n, 1),), -7, dtype=torch.int64, device=dev),
             reset_time=torch.full((max(n, 1),), -7, dtype=torch.int64, device=dev))
    b = ga.GuberBatch(n, 0, p["key_bytes"], p["key_off"], p["hits"], p["limit"], p["duration"], p["burst"], p["created_at"], p["algorithm"], p["behavior"],
                      p["is_owner"], None, None, hb.now_ms)
    res = ga.GuberResult(r["status"].data_ptr(), r["limit"].data_ptr(), r["remaining"].data_ptr(), r["reset_time"].data_ptr(), r["err"].data_ptr(), 0, 0, 0, 0, 0)
    return b, res, t, r


def check(hb, r, want, label):
    got = ga.HostResult(hb.n)
    for name in ("status", "limit", "remaining", "reset_time", "err"):
        getattr(got, name)[:hb.n] = r[name].cpu().numpy()[:hb.n]
    if hb.n:
        support.assert_results_equal(got, want, label)


@pytest.mark.parametrize("n_engines,n_streams,max_batch", [(4, 1, 8192), (12, 3, 8192), (6, 3, 2048), (1, 1, 8192)])
def test_generations_in_arrival_order_through_the_shards(n_engines, n_streams, max_batch):
    """twelve generations in two calls: keys of one width (they travel with their requests) and ragged ones (they stay in place), every
    request column or only the mandatory ones, an empty generation, generations of one and three requests, uniform keys; shares larger
    than an engine's max_batch go in pieces; hot keys are placed individually.  Every answer equals the oracle's, and so does the number
    of resident items"""
    import torch
    dev = torch.device("cuda", 0)
    K, G = 40_000, 32768
    tab = streams.key_table(K)
    place = ga.Placement(n_engines) if n_engines > 1 else None
    if place is not None:
        place.observe_keys(*streams.keys_for_ids(tab, streams.ZipfSampler(K, seed=77).draw(1 << 16)))
        place.rebalance(0.125, True)
    strs = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    engs = [ga.Engine(cache_size=1 << 18, max_batch=max_batch, stream=strs[j * n_streams // n_engines].cuda_stream) for j in range(n_engines)]
    fr = ga.Front(engs, place, max_n=G, depth=4)
    orc = Oracle(cache_size=1 << 20)
    zs = streams.ZipfSampler(K, seed=31)
    rng = np.random.default_rng(8)
    adv = streams.adversarial_batches(44, 6, 3000, greg_fn=support.gregorian)
    gens = []
    for g in range(12):
        now = streams.NOW0 + g * 700
        if g in (3, 7, 10):
            hb = next(adv)
            hb.behavior[:] = hb.behavior & ~np.uint32(4)
            hb.greg_expire[:] = 0
            hb.greg_duration[:] = 0
            gens.append((hb, True))
            continue
        n = [G, 20000, 0, G, 3, G, 7777, 6000, G, 1, G, 300][g]
        ids = rng.integers(0, K, n) if g in (5, 8) else zs.draw(n)
        hb = streams.bench_batch(tab, ids, now, algorithm=g % 2, limit=30, duration=4000)
        if n:
            hb.algorithm[:] = (np.arange(n) // 97 + g) % 2
        gens.append((hb, False))
    for lo, hi in ((0, 5), (5, 12)):
        part = [dev_gen(torch, dev, hb, full) for hb, full in gens[lo:hi]]
        torch.cuda.synchronize(dev)
        N = hi - lo
        assert fr.eval_dev((ga.GuberBatch * N)(*[x[0] for x in part]), (ga.GuberResult * N)(*[x[1] for x in part]), N) == N
        fr.synchronize()
        for k, (hb, _) in enumerate(gens[lo:hi]):
            check(hb, part[k][3], orc.eval(hb), f"generation {lo + k}")
    st = fr.stats()
    assert st["generations"] == 12 and st["forced_flushes"] == 0, st
    sizes = [e.size() for e in engs]
    assert sum(sizes) == orc.size() and min(sizes) > 0, (sizes, orc.size())
    if n_engines > 1 and max_batch >= 8192:
        assert sum(e.stats()["fused_batches"] for e in engs) > 0
    fr.close()
    for e in engs:
        e.close()
    if place is not None:
        place.close()


def test_a_long_stream_of_generations_keeps_every_keys_order():
    """64 generations of 65 536 requests over 200 000 keys on twelve engines over three streams, Zipf-1.1 (the hot key is a tenth of every
    generation), the clock stepping, in ONE call: the routing runs ahead of the evaluation, slots are reused, the k_eval3 of a generation
    rides on the next one's k_part — element-wise equal to the oracle"""
    import torch
    dev = torch.device("cuda", 0)
    K, G, NG, S = 200_000, 65536, 64, 12
    tab = streams.key_table(K)
    place = ga.Placement(S)
    place.observe_keys(*streams.keys_for_ids(tab, streams.ZipfSampler(K, seed=5).draw(1 << 18)))
    place.rebalance(0.125, True)
    strs = [torch.cuda.Stream(device=dev) for _ in range(3)]
    engs = [ga.Engine(cache_size=1 << 19, max_batch=65536, stream=strs[j * 3 // S].cuda_stream) for j in range(S)]
    fr = ga.Front(engs, place, max_n=G, depth=4)
    orc = Oracle(cache_size=1 << 21, workers=8)
    zs = streams.ZipfSampler(K, seed=1234)
    hbs = [streams.bench_batch(tab, zs.draw(G), streams.NOW0 + g * 37, algorithm=g % 2, limit=50, duration=1500) for g in range(NG)]
    part = [dev_gen(torch, dev, hb, False) for hb in hbs]
    torch.cuda.synchronize(dev)
    assert fr.eval_dev((ga.GuberBatch * NG)(*[x[0] for x in part]), (ga.GuberResult * NG)(*[x[1] for x in part]), NG) == NG
    fr.synchronize()
    for g, hb in enumerate(hbs):
        check(hb, part[g][3], orc.eval(hb, threads=8), f"generation {g}")
    st = fr.stats()
    assert st["generations"] == NG and st["forced_flushes"] == 0, st
    assert sum(e.stats()["retries"] for e in engs) == 0
    assert sum(e.size() for e in engs) == orc.size()
    fr.close()
    for e in engs:
        e.close()
    place.close()


@pytest.mark.parametrize("workers", [4, 3])
def test_the_front_over_binding_caches_is_the_references_worker_pool(workers):
    """The reference shards its cache over Config.Workers goroutines by the XXH64 of the HashKey (workers.go:125-151: CacheSize / Workers
    items each; getWorker :180-184) and every worker evicts in its own list's order (lrucache.go:88-149).  A front over `workers` engines
    whose placement is untouched (its initial table IS getWorker, also for worker counts that do not divide 2^63) with cache_size / workers
    items each must therefore answer a stream in arrival order exactly like the oracle with that many workers — through device routing,
    shares, eviction pre-passes per table (the caches bind: 2 600 keys over 2 000 items), requests that change a list's length, and the
    answers' way home; sizes and unexpired evictions included."""
    import torch
    dev = torch.device("cuda", 0)
    cs, nkeys, G = 2000, 2600, 4096
    place = ga.Placement(workers) if workers > 1 else None
    stream = torch.cuda.Stream(device=dev)
    engs = [ga.Engine(cache_size=cs // workers, max_batch=4096, stream=stream.cuda_stream) for _ in range(workers)]
    fr = ga.Front(engs, place, max_n=G, depth=4)
    orc = Oracle(cache_size=cs, workers=workers)
    gens = list(streams.length_changing_batches(29, 10, nkeys, 3000, "reset+greg", support.gregorian))
    for step, hb in enumerate(gens):
        hb.greg_expire[:] = 0                                   # (a front takes its calendar intervals from the device: the engines compute them in UTC,
        want = orc.eval(hb)                                      #  as support.gregorian does; invalid constants fail on either side)
        b, res, t, r = dev_gen(torch, dev, hb, True)
        torch.cuda.synchronize(dev)
        assert fr.eval_dev((ga.GuberBatch * 1)(b), (ga.GuberResult * 1)(res), 1) == 1
        fr.synchronize()
        check(hb, r, want, f"generation {step}")
        assert sum(e.size() for e in engs) == orc.size(), (step, [e.size() for e in engs], orc.size())
    assert sum(e.stats()["unexpired_evictions"] for e in engs) == orc.counters()[3]
    assert sum(e.stats()["eviction_passes"] for e in engs) >= 1
    fr.close()
    for e in engs:
        e.close()
    if place is not None:
        place.close()


def test_global_requests_go_to_the_global_engine():
    """guber_route_rule_t.global_engine (the pool keeps a device's GLOBAL keys in an engine of its own): requests that carry
    Behavior_GLOBAL land there whatever their key hashes to, the others follow the placement; answers equal ONE oracle (a key is either always
    GLOBAL or never in this stream)"""
    import torch
    dev = torch.device("cuda", 0)
    K, G = 30_000, 32768
    tab = streams.key_table(K)
    place = ga.Placement(2)
    stream = torch.cuda.Stream(device=dev)
    engs = [ga.Engine(cache_size=1 << 17, max_batch=32768, stream=stream.cuda_stream) for _ in range(3)]
    fr = ga.Front(engs, place, max_n=G, depth=4, global_engine=2)
    orc = Oracle(cache_size=1 << 20)
    zs = streams.ZipfSampler(K, seed=41)
    seen = []
    for g in range(5):
        ids = zs.draw(G)
        seen.append(ids)
        hb = streams.bench_batch(tab, ids, streams.NOW0 + g * 500, algorithm=g % 2, limit=40, duration=5000)
        hb.behavior[:] = np.where(ids % 3 == 0, 2, 0).astype(np.uint32)
        b, res, t, r = dev_gen(torch, dev, hb, False)
        torch.cuda.synchronize(dev)
        assert fr.eval_dev((ga.GuberBatch * 1)(b), (ga.GuberResult * 1)(res), 1) == 1
        fr.synchronize()
        check(hb, r, orc.eval(hb), f"generation {g}")
    u = np.unique(np.concatenate(seen))
    assert engs[2].size() == int((u % 3 == 0).sum())
    assert engs[0].size() + engs[1].size() == int((u % 3 != 0).sum()) and min(engs[0].size(), engs[1].size()) > 0
    fr.close()
    for e in engs:
        e.close()
    place.close()
