"""Pins the CPU oracle (oracle/guber_oracle.c) against every golden vector the reference's own tests
hold for the hot path (tests/golden/, transcribed from functional_test.go, store_test.go,
interval_test.go, replicated_hash_test.go, workers_internal_test.go) and against independent
implementations of the third-party hashes (python-xxhash, hashlib.md5)."""
import ctypes as C
import hashlib

import numpy as np
import pytest
import xxhash

import scenarios
import support
from support import Oracle, HostBatch, TOKEN, LEAKY, UNDER, OVER


def test_functional_vectors():
    n = scenarios.run_functional(lambda: Oracle())
    assert n >= 75


def test_functional_vectors_many_workers():
    # worker sharding (workers.go:180-184) must not be observable in results
    n = scenarios.run_functional(lambda: Oracle(cache_size=50000, workers=7))
    assert n >= 75


def test_store_vectors():
    assert scenarios.run_store(lambda: Oracle()) == 5


def test_gregorian_kats():
    lib = support.oracle_lib()
    k = scenarios.load("kat_vectors.json")
    for v in k["gregorian_expiration"]:
        out = C.c_int64(0)
        assert lib.oracle_gregorian_expiration(v["now_ns"], v["d"], C.byref(out)) == 0
        assert out.value == v["expire"], v
    inv = k["gregorian_invalid"]
    out = C.c_int64(7)
    rc = lib.oracle_gregorian_expiration(inv["now_ns"], inv["d"], C.byref(out))
    assert rc == -3 and out.value == 0
    assert support.ITEM_ERR_TEXT[3] == inv["error"]
    # interval.go:84-96 fixed durations; :93 weeks unsupported
    for d, want in [(0, 60000), (1, 3600000), (2, 86400000)]:
        assert lib.oracle_gregorian_duration(inv["now_ns"], d, C.byref(out)) == 0 and out.value == want
    assert lib.oracle_gregorian_duration(inv["now_ns"], 3, C.byref(out)) == -2


def test_ring_distribution_kat():
    lib = support.oracle_lib()
    k = scenarios.load("kat_vectors.json")["ring_distribution"]
    hosts = k["hosts"]
    arr = (C.c_char_p * len(hosts))(*[h.encode() for h in hosts])
    for kind, name in [(0, "fnv1"), (1, "fnv1a")]:
        ring = lib.oracle_ring_create(arr, len(hosts), k["replicas"], kind)
        dist = {h: 0 for h in hosts}
        for i in range(k["n_keys"]):
            ip = f"192.168.{(i >> 8) & 255}.{i & 255}".encode()
            dist[hosts[lib.oracle_ring_get(ring, ip, len(ip))]] += 1
        lib.oracle_ring_destroy(ring)
        assert dist == k[name], (name, dist)


def test_worker_index_kat():
    lib = support.oracle_lib()
    for v in scenarios.load("kat_vectors.json")["worker_index"]:
        assert lib.oracle_worker_index_for_hash63(v["workers"], v["hash63"]) == v["idx"]


def test_hashes_against_independent_implementations():
    lib = support.oracle_lib()
    rng = np.random.default_rng(5)
    for n in list(range(0, 70)) + [127, 128, 129, 1000]:
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for seed in (0, 1, 0x9E3779B185EBCA87):
            assert lib.oracle_xxhash64(b, n, seed) == xxhash.xxh64(b, seed=seed).intdigest(), (n, seed)
        d = C.create_string_buffer(16)
        lib.oracle_md5(b, n, d)
        assert d.raw == hashlib.md5(b).digest()
    # FNV-1 / FNV-1a 64 published test values
    assert lib.oracle_fnv1_64(b"", 0) == 0xcbf29ce484222325
    assert lib.oracle_fnv1_64(b"a", 1) == 0xaf63bd4c8601b7be
    assert lib.oracle_fnv1a_64(b"a", 1) == 0xaf63dc4c8601ec8c
    assert lib.oracle_fnv1a_64(b"foobar", 6) == 0x85944171f73967e8
    assert lib.oracle_fnv1_64(b"foobar", 6) == 0x340d8765a4dda9c2


def test_get_peer_rate_limits_order_stable():
    # functional_test.go:1638-1686: responses come back in request order, batch sizes 1..1000
    now = 1_700_000_000_000
    for n in [1, 2, 5, 10, 100, 1000]:
        o = Oracle()
        keys = [f"TestGetPeerRateLimits_k{n}_{i:05d}" for i in range(n)]
        res = o.eval(HostBatch(keys, 0, [1000 + i for i in range(n)], 1000, now))
        assert res.limit[:n].tolist() == [1000 + i for i in range(n)]
        assert (res.err[:n] == 0).all()


def test_invalid_algorithm_and_sequential_duplicates():
    o = Oracle()
    now = 1_700_000_000_000
    b = HostBatch(["a_1", "a_1", "a_1", "b_1"], 1, 2, 1000, now, algorithm=[0, 0, 0, 9])
    res = o.eval(b)
    assert res.rows()[0][:3] == (UNDER, 2, 1)
    assert res.rows()[1][:3] == (UNDER, 2, 0)
    assert res.rows()[2][:3] == (OVER, 2, 0)       # same key, applied in request order
    assert res.rows()[3] == (0, 0, 0, 0, 1)        # workers.go:318, nil response
    assert res.counters()[:3] == (1, 2, 1)         # over-limit, hits, misses


def test_lru_eviction_and_counters():
    # lrucache.go:98-100,138-149: evict the least recently used when over cacheSize
    o = Oracle(cache_size=3, workers=1)
    now = 1_700_000_000_000
    o.eval(HostBatch(["k_1", "k_2", "k_3"], 1, 10, 60000, now))
    assert o.size() == 3
    o.eval(HostBatch(["k_1"], 1, 10, 60000, now))             # touch k_1 -> k_2 is now oldest
    res = o.eval(HostBatch(["k_4"], 1, 10, 60000, now))
    assert o.size() == 3 and res.counters()[3] == 1           # unexpired eviction counted
    assert o.get_item("k_2", now) is None
    assert o.get_item("k_1", now)["remaining"] == 8
    # expired items are removed lazily on access (lrucache.go:115-119)
    assert o.get_item("k_1", now + 60001) is None


def test_mt_matches_sequential():
    rng = np.random.default_rng(11)
    now = 1_700_000_000_000
    a, b = Oracle(cache_size=1 << 20, workers=8), Oracle(cache_size=1 << 20, workers=8)
    for step in range(4):
        ids = rng.zipf(1.3, 5000) % 700
        keys = [f"mt_{i}" for i in ids]
        hb = HostBatch(keys, rng.integers(0, 4, 5000), 20, 50, now + step * 30,
                       algorithm=(ids % 2).astype(np.uint8))
        support.assert_results_equal(a.eval(hb, threads=4), b.eval(hb), f"step {step}")


def test_store_callbacks_follow_teststore():
    """store_test.go TestStore: Get on a miss, OnChange after the request, Remove on a foreign Value."""
    assert scenarios.run_store_events(lambda: Oracle(cache_size=1 << 12)) == 10


def test_lrucache_vectors():
    """lrucache_test.go TestLRUCache: Add / GetItem / Remove / Size, replace-on-Add, and which evictions are counted."""
    assert scenarios.run_cache_vectors(lambda cs: Oracle(cache_size=cs, workers=1)) > 3000
