"""The C++ host layer (GPUWorkerPool + V1Instance.GetRateLimits, csrc/worker_pool.h) driven the way the
reference's functional tests drive a daemon: a frozen clock, one GetRateLimits call per step."""
import os
import threading

import numpy as np
import pytest

import gubernator_amd as ga
import scenarios
import support

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shards", [1, 4])
def test_functional_vectors_through_v1instance(shards):
    """functional_test.go tables replayed through V1Instance.GetRateLimits (frozen clock = clock.Freeze)."""
    n = 0
    for sc in scenarios.load("functional_vectors.json")["scenarios"]:
        inst = ga.V1Instance(cache_size=4096, batch_limit=64, batch_wait_us=50, shards=shards)
        now = sc["start_ms"]
        for si, step in enumerate(sc.get("steps", [])):
            inst.set_clock(now)
            r = inst.GetRateLimits([step["req"]])[0]
            exp = step["expect"]
            where = f"{sc['name']} step {si}"
            if "error" in exp:
                assert r["error"] == exp["error"], where          # TestMissingFields strings (gubernator.go:208-217)
            if exp.get("error"):
                n += 1
                now += step["advance_ms"]
                continue
            assert r["error"] == "", where
            scenarios.check_expect(exp, (r["status"], r["limit"], r["remaining"], r["reset_time"], 0), now, where)
            n += 1
            now += step["advance_ms"]
        for step in sc.get("batch_steps", []):
            inst.set_clock(now)
            out = inst.GetRateLimits(step["reqs"])
            for j, exp in enumerate(step["expect"]):
                scenarios.check_expect(exp, (out[j]["status"], out[j]["limit"], out[j]["remaining"], out[j]["reset_time"], 0), now, sc["name"])
                n += 1
        inst.close()
    assert n >= 80


def test_request_list_too_large_and_error_texts():
    inst = ga.V1Instance(cache_size=4096)
    reqs = [dict(name="big", unique_key=f"k{i}", hits=1, limit=5, duration=1000) for i in range(1001)]
    with pytest.raises(ga.GuberError) as ei:                       # gubernator.go:189-193 codes.OutOfRange
        inst.GetRateLimits(reqs)
    assert "Requests.RateLimits list too large; max size is '1000'" in str(ei.value)
    out = inst.GetRateLimits(reqs[:1000])                           # exactly 1000 is accepted, order kept
    assert [o["remaining"] for o in out] == [4] * 1000
    bad = inst.GetRateLimits([dict(name="x", unique_key="y", hits=1, limit=5, duration=1000, algorithm=7),
                              dict(name="x", unique_key="g", hits=1, limit=5, duration=99, behavior=4),
                              dict(name="x", unique_key="w", hits=1, limit=5, duration=3, behavior=4)])
    assert bad[0]["error"] == "Error while apply rate limit for 'x_y': Invalid rate limit algorithm '7'"   # workers.go:318
    assert bad[1]["error"].endswith("behavior DURATION_IS_GREGORIAN is set; but `Duration` is not a valid gregorian interval")
    assert bad[2]["error"].endswith("`Duration = GregorianWeeks` not yet supported; consider making a PR!`")
    inst.close()


def test_peer_order_stability_batch_sizes():
    # functional_test.go:1638-1686: responses in request order for batch sizes 1..1000
    inst = ga.V1Instance(cache_size=8192)
    for n in [1, 2, 5, 10, 100, 1000]:
        reqs = [dict(name="TestGetPeerRateLimits", unique_key=f"{n}_{i}", hits=0, limit=1000 + i, duration=1000) for i in range(n)]
        assert [o["limit"] for o in inst.GetRateLimits(reqs)] == [1000 + i for i in range(n)]
    inst.close()


@pytest.mark.parametrize("shards", [1, 4])
def test_concurrent_callers_are_batched_and_consistent(shards):
    """Many goroutine-like callers on one key set (benchmark_test.go "Thundering herd" shape): every hit is
    accounted exactly once — admitted hits == limit per key — and callers share device batches."""
    import os
    os.environ["GUBER_POOL_EAGER"] = "0"          # the reference's peer-batcher policy alone (limit or wait): arrivals inside batch_wait share a batch
    try:
        inst = ga.V1Instance(cache_size=8192, batch_limit=512, batch_wait_us=300, shards=shards)
    finally:
        del os.environ["GUBER_POOL_EAGER"]
    inst.set_clock(1_700_000_000_000)
    keys, limit, threads, per_thread = 20, 50, 16, 25
    admitted = np.zeros(keys, np.int64)
    lock = threading.Lock()

    def worker(t):
        local = np.zeros(keys, np.int64)
        for j in range(per_thread):
            reqs = [dict(name="herd", unique_key=f"k{k}", hits=1, limit=limit, duration=600_000) for k in range(keys)]
            for k, o in enumerate(inst.GetRateLimits(reqs)):
                assert o["error"] == ""
                local[k] += 1 if o["status"] == 0 else 0
        with lock:
            admitted[:] += local
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert (admitted == limit).all(), admitted                     # 16*25 = 400 hits per key, limit 50
    assert inst.batches() < threads * per_thread * shards           # callers were coalesced into shared batches
    inst.close()


def test_golden_functional_scenarios_through_the_wire_on_the_gpu():
    """serialized GetRateLimitsReq -> guber_wire_decode_requests -> guber_wire_eval (HIP engine) ->
    guber_wire_encode_responses -> parsed by the python protobuf runtime -> the reference's expectations."""
    import wire_replay
    from gubernator_amd import wire as gw

    def make_eval():
        e = ga.Engine(cache_size=4096, max_batch=2048)
        return (lambda wb: wb.eval(e)), e.close
    assert wire_replay.run_functional_wire(lambda: gw.WireBatch(2048, 1 << 16, pinned=True), make_eval) >= 80


def test_aggregated_rpc_payloads_match_the_oracle_on_the_gpu():
    """70 RPC payloads of up to 1000 items (the reference's per-RPC cap) aggregated into ONE device batch; every
    RPC's response slice must equal what the oracle answers for the same aggregated batch."""
    import ctypes as C
    import wire_replay
    from gubernator_amd import wire as gw
    import support
    from pb_schema import PB
    rng = np.random.default_rng(21)
    now = 1_700_000_000_000
    e = ga.Engine(cache_size=1 << 16, max_batch=65536)
    o = support.Oracle(cache_size=1 << 17)
    gpu, cpu = gw.WireBatch(65536, 4 << 20, pinned=True), gw.WireBatch(65536, 4 << 20)
    for rounds in range(3):
        gpu.reset(now); cpu.reset(now)
        slices = []
        for rpc in range(70):
            n = int(rng.integers(1, 1001))
            reqs = [dict(name="agg", unique_key="k%d" % int(rng.zipf(1.3)) if rng.random() > 0.01 else "", hits=int(rng.integers(0, 3)),
                         limit=50, duration=int(rng.choice([1, 60_000])), algorithm=int(rng.integers(0, 3) % 3 if rng.random() < 0.02 else rng.integers(0, 2)),
                         behavior=int(rng.choice([0, 0, 0, 8, 32])), burst=0) for _ in range(n)]
            p = wire_replay.pb_request(reqs)
            try:
                s = gpu.decode(p, max_per_rpc=1000)
            except ga.GuberError as ex:
                assert ex.code == gw.E_WIRE_FULL
                break
            assert cpu.decode(p, max_per_rpc=1000) == s
            slices.append(s)
        gpu.eval(e)
        o.lib.oracle_eval_batch(o.h, C.byref(cpu.view()), C.byref(cpu.result()))
        for first, count in slices:
            a, b = gpu.encode(first, count), cpu.encode(first, count)
            assert a == b, (rounds, first, count)
            m = PB["GetRateLimitsResp"]()
            m.ParseFromString(a)
            assert len(m.responses) == count
        now += 7
    e.close()


@pytest.mark.parametrize("shards", [1, 3])
def test_store_callbacks_through_the_pool_teststore(shards):
    """store_test.go TestStore through the C++ pool: Config.Store as C callbacks (guber_pool_set_store); the batcher
    must call Get on the miss, Remove for a foreign Value, OnChange with the item after the request.  With several shards the
    requests sit in the device's front stage in arrival order and every shard's share goes through the Store sequence on its own
    engine (worker_pool.cpp submit_with_store)."""
    import scenarios
    import support

    class PoolBackend:
        def __init__(self):
            self.inst = ga.V1Instance(cache_size=4096, batch_limit=64, batch_wait_us=200, shards=shards)

        def eval_store(self, batch, store):
            # adapter: MockStore speaks (req_index, key); the pool hands (req dict, key)
            class A:
                def get(_s, r, key): return store.get(0, key)
                def on_change(_s, r, key, item): store.on_change(0, key, item)
                def remove(_s, r, key): store.remove(0, key)
            self.inst.set_clock(batch.now_ms)
            self.inst.set_store(A())
            keys = [support.batch_key(batch, i) for i in range(batch.n)]
            reqs = [dict(name=k.split("_account")[0], unique_key="account" + k.split("_account")[1], hits=int(batch.hits[i]),
                         limit=int(batch.limit[i]), duration=int(batch.duration[i]), algorithm=int(batch.algorithm[i]),
                         behavior=int(batch.behavior[i]), burst=int(batch.burst[i])) for i, k in enumerate(keys)]
            out = self.inst.GetRateLimits(reqs)
            res = support.HostResult(batch.n)
            for i, o in enumerate(out):
                assert o["error"] == ""
                res.status[i], res.limit[i], res.remaining[i], res.reset_time[i], res.err[i] = o["status"], o["limit"], o["remaining"], o["reset_time"], 0
            return res

        def close(self):
            self.inst.close()
    assert scenarios.run_store_events(PoolBackend) == 10


@pytest.mark.parametrize("shards", [1, 4])
def test_store_is_asked_again_after_a_reset_inside_one_batch(shards):
    """algorithms.go:45-51 after :78-90 — RESET_REMAINING removes the item from cache and store; the key's next request, in the SAME
    GetRateLimits call, misses the cache again and the reference calls Store.Get again.  A store that still has the item (it
    ignores Remove) shows the difference: the third request continues from the store's state, not from a fresh bucket."""
    now = 1_700_000_000_000
    calls = []

    class Sticky:
        def get(self, r, key):
            calls.append(("get", key))
            return dict(algorithm=0, limit=10, duration=60_000, remaining=7, stamp=now, expire_at=now + 60_000)

        def on_change(self, r, key, item):
            calls.append(("on_change", key, item["remaining"]))

        def remove(self, r, key):
            calls.append(("remove", key))
    inst = ga.V1Instance(cache_size=4096, batch_limit=64, batch_wait_us=200, shards=shards)
    inst.set_clock(now)
    inst.set_store(Sticky())
    req = dict(name="sticky", unique_key="account:1", hits=1, limit=10, duration=60_000)
    out = inst.GetRateLimits([req, dict(req, behavior=8), req, dict(name="sticky", unique_key="account:2", hits=1, limit=10, duration=60_000), req])
    key = "sticky_account:1"
    assert [o["remaining"] for o in out] == [6, 10, 6, 6, 5], out            # 7-1; reset answer (Limit); store's 7 again -1; other key 7-1; 6-1
    assert out[1]["reset_time"] == 0                                          # algorithms.go:84-89
    mine = [c for c in calls if c[1] == key]
    assert mine == [("get", key), ("on_change", key, 6), ("remove", key), ("get", key), ("on_change", key, 6), ("on_change", key, 5)], mine
    inst.close()


@pytest.mark.parametrize("shards", [1, 3])
def test_loader_round_trip_through_the_pool(shards):
    """store_test.go:76-125 TestLoader: items handed over by Loader.Load are served from the cache, and Loader.Save at
    shutdown receives every resident item with its current state ({Limit 2, Remaining 1, UNDER} after one hit)."""
    now = 1_700_000_000_000
    inst = ga.V1Instance(cache_size=4096, batch_limit=64, batch_wait_us=200, shards=shards)
    inst.set_clock(now)
    inst.load([ga.make_item(f"loaded_k{i}", 0, limit=10, duration=60_000, remaining=10 - i % 5, stamp=now - 5, expire_at=now + 59_995)
               for i in range(300)])
    out = inst.GetRateLimits([dict(name="test_over_limit", unique_key="account:1234", hits=1, limit=2, duration=1000),
                              dict(name="loaded", unique_key="k3", hits=1, limit=10, duration=60_000)])
    assert (out[0]["status"], out[0]["remaining"]) == (0, 1)
    assert (out[1]["status"], out[1]["remaining"]) == (0, 10 - 3 - 1)                   # continued from the loaded state
    saved = {d["key"]: d for d in inst.store()}
    assert len(saved) == 301
    it = saved[b"test_over_limit_account:1234"]
    assert (it["algorithm"], it["limit"], it["remaining"], it["status"]) == (0, 2, 1, 0)
    assert saved[b"loaded_k3"]["remaining"] == 6 and saved[b"loaded_k7"]["remaining"] == 10 - 7 % 5
    inst.close()


@pytest.mark.parametrize("shards", [1, 4])
def test_a_pool_whose_caches_bind_evicts_like_the_reference_workers(shards):
    """workers.go:125-140: every worker has an LRUCache of CacheSize / workers items (lrucache.go:88-149).  A pool over a key
    population larger than that, 1 000-item RPCs whose evicted keys come back at once: every answer equals the reference's worker
    pool (the oracle with the same worker rule and per-worker bounded LRU) — through the pool's front stage, whose shares then go
    through the engines' eviction pre-pass one engine at a time (guber_stage_submit_routed)."""
    os.environ["GUBER_POOL_REBALANCE_MS"] = "0"          # the placement stays the reference's worker rule (no hot-key moves)
    try:
        inst = ga.V1Instance(cache_size=1200, batch_limit=1000, batch_wait_us=50, shards=shards)
    finally:
        del os.environ["GUBER_POOL_REBALANCE_MS"]
    orc = support.Oracle(cache_size=1200, workers=shards)
    rng = np.random.default_rng(21)
    now = 1_700_000_000_000
    for step in range(20):
        inst.set_clock(now)
        ids = rng.integers(0, 1700, 1000) if step % 3 else (step * 700 + np.arange(1000)) % 1700
        reqs = [dict(name="lru", unique_key=f"k{int(i)}", hits=1, limit=1000, duration=3_600_000) for i in ids]
        out = inst.GetRateLimits(reqs)
        hb = support.HostBatch([f"lru_k{int(i)}" for i in ids], 1, 1000, 3_600_000, now, created_at=now)
        want = orc.eval(hb)
        got = np.array([o["remaining"] for o in out])
        assert all(o["error"] == "" for o in out)
        assert np.array_equal(got, np.asarray(want.remaining[:hb.n])), (step, np.nonzero(got != np.asarray(want.remaining[:hb.n]))[0][:10])
        now += 100
    assert inst.size() == orc.size()
    inst.close()


def test_pool_shards_follow_the_reference_worker_rule():
    """WorkerPool.getWorker (workers.go:153-155,180-184; workers_internal_test.go:51-54): shard = XXH64(key) >> 1 divided by
    2^63 / workers — the pool's key -> shard map must be exactly that, and every key must live in exactly one shard."""
    import ctypes as C
    import xxhash
    inst = ga.V1Instance(cache_size=4096, batch_limit=64, batch_wait_us=100, shards=5)
    L = ga.lib()
    L.guber_pool_shard_of.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
    L.guber_pool_shard_of.restype = C.c_uint32
    o = support.oracle_lib()
    seen = set()
    for i in range(2000):
        k = b"shardkey_%d" % i
        want = o.oracle_worker_index_for_hash63(5, xxhash.xxh64(k, seed=0).intdigest() >> 1)
        assert L.guber_pool_shard_of(inst.h, k, len(k)) == want
        seen.add(want)
    assert seen == {0, 1, 2, 3, 4}
    inst.set_clock(1_700_000_000_000)
    out = inst.GetRateLimits([dict(name="shardkey", unique_key=str(i), hits=1, limit=3, duration=60_000) for i in range(500)] * 2)
    assert [o_["remaining"] for o_ in out[:500]] == [2] * 500 and [o_["remaining"] for o_ in out[500:]] == [1] * 500
    assert len(inst.store()) == 500
    inst.close()


@pytest.mark.parametrize("workers", [1, 4])
def test_gubernator_pool_load_store(workers):
    """workers_test.go:31-130 TestGubernatorPool (Single-threaded / Multi-threaded): Load() hands the Loader's 100 items to the
    workers' caches, Store() gives Loader.Save exactly those items back (compared sorted by key, as the reference does)."""
    inst = ga.V1Instance(cache_size=4096, batch_limit=64, batch_wait_us=200, shards=workers)
    inst.set_clock(1_700_000_000_000)
    loaded = [dict(key=b"Foobar%04d" % i, algorithm=0, limit=10 + i, duration=1000, remaining=i, stamp=1_700_000_000_000 - i,
                   expire_at=4131978658000) for i in range(100)]
    inst.load([ga.make_item(d["key"], d["algorithm"], limit=d["limit"], duration=d["duration"], remaining=d["remaining"], stamp=d["stamp"],
                            expire_at=d["expire_at"]) for d in loaded])
    saved = sorted(inst.store(), key=lambda d: d["key"])
    assert len(saved) == 100
    for want, got in zip(loaded, saved):
        for f in ("key", "algorithm", "limit", "duration", "remaining", "stamp", "expire_at"):
            assert got[f] == want[f], (f, want, got)
    inst.close()


def test_multi_device_pool_routes_by_the_ring_and_answers_in_place():
    """guber_pool_create_multi with three LOGICAL devices on this GPU (peers gpu0..gpu2 of the replicated consistent hash) x 2
    shards: a key's device is the ring's owner (replicated_hash.go:104-119), mixed batches are split by owner and every request
    is answered in its own slot (functional_test.go:1638-1686) — equal to ONE unsharded oracle on a mixed stream; the batcher
    metrics move (gubernator.go:96-107 analogues); over-long keys are answered per item without touching the device."""
    from support import HostBatch, Oracle
    inst = ga.V1Instance(cache_size=60_000, batch_limit=512, batch_wait_us=100, shards=2, devices=[0, 0, 0], max_key_bytes=64)
    assert inst.n_shards() == 6
    ring = ga.Ring(["gpu0", "gpu1", "gpu2"])
    keys = [f"multi_k{i}" for i in range(3000)]
    want_dev = ring.route([f"mdp_{k}" for k in keys])
    assert [inst.device_of(f"mdp_{k}") for k in keys] == want_dev.tolist()
    assert all(inst.shard_of(f"mdp_{k}") // 2 == d for k, d in zip(keys, want_dev.tolist()))
    o = Oracle(cache_size=1 << 20)
    rng = np.random.default_rng(8)
    now = 1_700_000_000_000
    for step in range(30):
        inst.set_clock(now)
        ids = rng.integers(0, 300, 900)
        reqs = [dict(name="mdp", unique_key=f"multi_k{int(i)}", hits=int(h), limit=40, duration=60_000, algorithm=int(i) % 2, created_at=now)
                for i, h in zip(ids, rng.choice([0, 1, 2], 900))]
        out = inst.GetRateLimits(reqs)
        want = o.eval(HostBatch([f"mdp_multi_k{int(i)}" for i in ids], [r["hits"] for r in reqs], 40, 60_000, now, algorithm=(ids % 2).astype(np.uint8),
                                created_at=now))
        got_rows = [(x["status"], x["limit"], x["remaining"], x["reset_time"], 0 if not x["error"] else 1) for x in out]
        assert got_rows == want.rows(), f"step {step}"
        now += 7
    sizes = [inst.shard_size(j) for j in range(6)]
    # (fnv1 — the reference's library default — spreads keys that differ only in their last digits unevenly: no balance claim here)
    assert sum(sizes) == o.size() and sum(1 for d in range(3) if sizes[2 * d] + sizes[2 * d + 1] > 0) >= 2, sizes
    m = inst.metrics()
    assert m["devices"] == 3 and m["shards"] == 6 and m["requests"] == 30 * 900 and m["batches"] >= 30 and m["queue_length"] == 0
    assert m["send_duration_us_sum"] > 0 and m["batch_size_max"] <= 512 * 2 and m["in_flight"] == 0       # (a device's front stage: batch_limit per shard)
    long_key = "x" * 80
    out = inst.GetRateLimits([dict(name="mdp", unique_key=long_key, hits=1, limit=5, duration=1000), dict(name="mdp", unique_key="ok", hits=1, limit=5, duration=1000)])
    assert "too long" in out[0]["error"] and out[1]["error"] == "" and out[1]["remaining"] == 4
    assert inst.metrics()["key_too_long"] == 1
    inst.close()


def test_pool_global_engine_and_hot_key_migration():
    """(a) With GUBER_FLAG_GLOBAL the pool keeps the keys of GLOBAL requests in ONE dedicated engine per device (the replica the
    GLOBAL manager synchronises) and guber_pool_global_sync ticks natively over the devices: the owner's state is installed on the
    other device's replica (global.go:234-283).  (b) A key that carries most of the traffic is moved to another logical shard WITH
    its bucket by the dispatcher's placement pass, and nothing a caller sees changes: equal to ONE unsharded oracle throughout."""
    from support import HostBatch, Oracle
    now = 1_700_000_000_000
    inst = ga.V1Instance(cache_size=40_000, batch_limit=512, batch_wait_us=100, shards=4, devices=[0, 0], flags=ga.FLAG_GLOBAL, max_key_bytes=64)
    inst.set_clock(now)
    assert inst.n_shards() == 2 * (4 + 1)
    o = Oracle(cache_size=1 << 20)
    greqs = [dict(name="glob", unique_key=f"g{i}", hits=1, limit=50, duration=60_000, behavior=2, created_at=now) for i in range(200)]
    out = inst.GetRateLimits(greqs)
    want = o.eval(HostBatch([f"glob_g{i}" for i in range(200)], 1, 50, 60_000, now, behavior=2, created_at=now))
    assert [(x["status"], x["limit"], x["remaining"], x["reset_time"]) for x in out] == [r[:4] for r in want.rows()]
    assert inst.global_engine_size(0) + inst.global_engine_size(1) == 200 == inst.size()      # only the GLOBAL engines hold them
    st = inst.global_sync()
    assert st["update_rows"] == 200 and st["items_installed"] == 200 and st["fallbacks"] == 0, st   # every owner broadcast, the other replica installed
    assert inst.global_engine_size(0) == inst.global_engine_size(1) == 200
    # (a2) UpdatePeerGlobals (gubernator.go:425-459) installs broadcast GLOBAL state through AddCacheItem: it must land where the
    # non-owner's GLOBAL requests are answered from — the device's GLOBAL engine — and GetCacheItem must find it there
    import support
    it = support.make_item("glob_peer0", 0, limit=50, duration=60_000, remaining=7, stamp=now, expire_at=now + 60_000, status=0)
    inst.add_item(it, behavior=2)
    dv = inst.device_of("glob_peer0")
    assert inst.global_engine_size(dv) == 201 and inst.size() == 401
    got = inst.get_item("glob_peer0")
    assert got is not None and got["remaining"] == 7 and got["limit"] == 50
    out = inst.GetRateLimits([dict(name="glob", unique_key="peer0", hits=0, limit=50, duration=60_000, behavior=2, created_at=now)], is_owner=[False])
    assert (out[0]["status"], out[0]["remaining"], out[0]["error"]) == (0, 7, ""), out
    # without a behaviour the item goes where its key already lives: the GLOBAL engine for this key, a plain shard for a new one
    it2 = support.make_item("glob_peer0", 0, limit=50, duration=60_000, remaining=3, stamp=now, expire_at=now + 60_000, status=0)
    inst.add_item(it2)
    assert inst.global_engine_size(dv) == 201 and inst.get_item("glob_peer0")["remaining"] == 3
    inst.add_item(support.make_item("plain_x", 0, limit=9, duration=60_000, remaining=4, stamp=now, expire_at=now + 60_000, status=0))
    assert inst.global_engine_size(0) + inst.global_engine_size(1) == 401 and inst.size() == 402 and inst.get_item("plain_x")["remaining"] == 4
    out = inst.GetRateLimits([dict(name="plain", unique_key="x", hits=1, limit=9, duration=60_000, created_at=now)])
    assert (out[0]["status"], out[0]["remaining"]) == (0, 3), out
    extra = 2                                                       # items added above (glob_peer0, plain_x), on top of the oracle's
    # (b) one key hammered, placement passes asked for in between
    rng = np.random.default_rng(5)
    L = ga.lib()
    import ctypes
    L.guber_pool_rebalance.argtypes = [ctypes.c_void_p]
    L.guber_pool_rebalance.restype = None
    for step in range(30):
        ids = np.where(rng.random(900) < 0.6, 7, rng.integers(0, 400, 900))
        reqs = [dict(name="hot", unique_key=f"k{int(i)}", hits=1, limit=100_000, duration=600_000, algorithm=int(i) % 2, created_at=now) for i in ids]
        got = inst.GetRateLimits(reqs)
        want = o.eval(HostBatch([f"hot_k{int(i)}" for i in ids], 1, 100_000, 600_000, now, algorithm=(ids % 2).astype(np.uint8), created_at=now))
        assert [(x["status"], x["limit"], x["remaining"], x["reset_time"], 0 if not x["error"] else 1) for x in got] == want.rows(), step
        L.guber_pool_rebalance(inst.h)
    m = inst.metrics()
    assert m["rebalances"] >= 1 and m["keys_moved"] >= 1, m
    assert inst.size() == o.size() + 200 + extra                            # (the GLOBAL keys live on both devices' replicas)
    inst.close()


def test_the_go_bindings_call_sequence_in_plain_c(tmp_path):
    """tests/hostsim/abi_c99.c — the cgo preamble and the calls go/gpu_worker_pool.go makes, compiled as C99 — against the real
    library on the GPU: create, GetRateLimits with owner flags, AddCacheItem (GLOBAL), GetCacheItem, Load, Store, GlobalSync, Close; then
    the front over two tables, and go/wire_server.go's calls: a serialized GetRateLimitsReq through the payload stage, the response bytes checked."""
    import subprocess
    exe = str(tmp_path / "abi_c99")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "gubernator_amd")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"), "-o", exe,
                    os.path.join(root, "tests", "hostsim", "abi_c99.c"), "-L", libdir, "-lguber_hip", f"-Wl,-rpath,{libdir}"], check=True)
    r = subprocess.run([exe, "--gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
