"""Parity tests proper: the HIP path, called through the C ABI (include/guber_gpu.h), against the
CPU oracle on the same seeded inputs — bit-exact on status / limit / remaining / reset_time / err,
for token AND leaky buckets (the leaky float64 math is IEEE division, add, subtract and conversions
only, so the tolerance is 0)."""
import os

import numpy as np
import pytest

import gubernator_amd as ga
import scenarios
import streams
import support
from support import HostBatch, Oracle

pytestmark = pytest.mark.gpu


def engine(**kw):
    kw.setdefault("cache_size", 1 << 16)
    kw.setdefault("max_batch", 65536)
    return ga.Engine(**kw)


def test_golden_functional_vectors():
    assert scenarios.run_functional(lambda: engine(cache_size=4096, max_batch=1024)) >= 75                               # one-launch small path
    assert scenarios.run_functional(lambda: engine(cache_size=4096, max_batch=1024, flags=ga.FLAG_TEST_NO_SMALL)) >= 75   # two-launch pipeline


def test_golden_store_vectors():
    assert scenarios.run_store(lambda: engine(cache_size=4096, max_batch=1024)) == 5


def test_get_peer_rate_limits_order_stable():
    # functional_test.go:1638-1686
    now = streams.NOW0
    for n in [1, 2, 5, 10, 100, 1000]:
        e = engine(cache_size=4096, max_batch=1024)
        keys = [f"TestGetPeerRateLimits_k{n}_{i:05d}" for i in range(n)]
        res = e.eval(HostBatch(keys, 0, [1000 + i for i in range(n)], 1000, now))
        assert res.limit[:n].tolist() == [1000 + i for i in range(n)]
        assert (res.err[:n] == 0).all()
        e.close()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("nkeys", [513, 700])
def test_more_keys_of_one_owner_than_its_lds_table_has_cells(nkeys):
    """the owner-partitioned pipeline, 513 / 700 DISTINCT keys that share ONE owner workgroup (the top 8 bits of the home position) in
    one batch: one round's worth of messages, more keys than the round's LDS hash table has cells.  k_own's insert loop is bounded
    and the round splits (it used to probe the full table for ever: tests/test_kernels_devsim.py has the CPU twin)"""
    slots = 1 << 20
    ol = support.oracle_lib()
    keys, i = [], 0
    while len(keys) < nkeys:
        k = b"own_%d" % i
        i += 1
        if ((ol.oracle_xxhash64(k, len(k), 0) >> 7) & (slots - 1)) >> 12 == 0:
            keys.append(k)
    o, e = Oracle(cache_size=1 << 20), engine(cache_size=1 << 16, max_batch=4096, table_slots=slots, flags=ga.FLAG_TEST_FORCE_PART)
    assert e.stats()["table_slots"] == slots
    for rnd in range(2):
        b = HostBatch(keys, 1, 5, 60000, streams.NOW0 + rnd)
        support.assert_results_equal(e.eval(b), o.eval(b), f"round {rnd}")
    e.close()


@pytest.mark.parametrize("bits", ["7", "8", ""])
def test_owner_partitioned_pipeline_with_either_owner_count(bits, monkeypatch):
    """the owner-partitioned pipeline splits a batch over 128 or 256 owner workgroups, following the traffic on the device
    (guber_kernels_part.h "HOW MANY OWNERS"; GUBER_PT_BITS pins it): adversarial streams, then batches of 40 000 distinct keys
    (312 / 156 per owner: with 128 owners some rounds split, the count moves to 256 by itself when it is free to) — equal to the oracle"""
    if bits:
        monkeypatch.setenv("GUBER_PT_BITS", bits)
    else:
        monkeypatch.delenv("GUBER_PT_BITS", raising=False)
    o, e = Oracle(cache_size=1 << 20), engine(cache_size=1 << 18, flags=ga.FLAG_TEST_FORCE_PART)
    for bi, b in enumerate(streams.adversarial_batches(5, 12, 3000, greg_fn=support.gregorian)):
        support.assert_results_equal(e.eval(b), o.eval(b), f"bits {bits!r} batch {bi}")
    table = streams.key_table(120_000)
    rng = np.random.default_rng(17)
    for rnd in range(4):
        ids = rng.permutation(120_000)[:40_000]
        b = streams.bench_batch(table, ids, streams.NOW0 + 100 + rnd, limit=20, duration=60_000)
        support.assert_results_equal(e.eval(b), o.eval(b), f"bits {bits!r} distinct keys {rnd}")
    e.close()


def test_k_eval3_and_the_next_k_part_share_a_launch():
    """the engine's default for routed calls (k_evalpart_multi: two launches per pass instead of three): four and six tables on one
    stream, 30 rounds of adversarial Zipf batches incl. a batch too small for the pipeline, a round without one table, uniform keys
    that move the owner count — every batch equal to its table's oracle, sizes too, and the fused launch really carried the passes
    (tests/fuse_ep_check.py, in a process of its own: the engines read GUBER_FUSE_EP once)"""
    import subprocess, sys
    env = {k: v for k, v in os.environ.items() if k != "GUBER_FUSE_EP"}
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuse_ep_check.py")], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and "FUSE_EP CHECK OK" in p.stdout, (p.stdout[-2000:], p.stderr[-2000:])
    assert "k_evalpart_multi" in p.stdout


# flags 0 = two-launch tile-bitmap pipeline (batches <= 65536), claims in the engine's claim table; 4 = the same in careful
# mode (verify first, claims keyed by the bucket slot: the retry round's code path); 2 = force the large-batch radix pipeline;
# 32 = batches of <= 256 requests through the two-launch pipeline as well (with 0 they take the one-launch small path);
# 64 = every batch through the owner-partitioned three-launch pipeline (k_part / k_own / k_eval3; what device-resident batches of
# >= 1024 requests take by default), 128 = never that one (the two-launch pipeline with per-batch claims)
@pytest.mark.parametrize("flags", [0, 2, 4, 32, 64, 128])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_adversarial_streams(seed, flags):
    o, e = Oracle(cache_size=1 << 20), engine(flags=flags)
    for bi, b in enumerate(streams.adversarial_batches(seed, 60, 3000, greg_fn=support.gregorian)):
        want, got = o.eval(b), e.eval(b)
        support.assert_results_equal(got, want, f"seed {seed} batch {bi}")
        assert got.counters() == want.counters(), f"counters seed {seed} batch {bi}: {got.counters()} {want.counters()}"
    e.close()


@pytest.mark.parametrize("flags", [0, 2, 4, 64])
def test_hot_key_runs(flags):
    now = streams.NOW0
    for algo in (0, 1):
        for beh in (0, 32):
            for hits, limit in [(1, 100), (3, 100), (1, 5000), (5, 5)]:
                o, e = Oracle(cache_size=1 << 16), engine(cache_size=1024, max_batch=8192, flags=flags)
                for step in range(3):
                    b = HostBatch([b"hot_key"] * 6000, hits, limit, 60_000, now + step * 1700, algorithm=algo, behavior=beh)
                    support.assert_results_equal(e.eval(b), o.eval(b), f"algo {algo} beh {beh} hits {hits} step {step}")
                e.close()


@pytest.mark.parametrize("flags", [0, 2, 64])
def test_edge_cases_empty_ragged_long_keys(flags):
    o, e = Oracle(cache_size=1 << 16), engine(cache_size=4096, max_batch=4096, max_key_bytes=300, flags=flags)
    now = streams.NOW0
    # empty batch
    res = e.eval(HostBatch([], [], [], [], now))
    assert res.n == 0
    # ragged key lengths 1..300 (inline <= 62 bytes, arena beyond), duplicates of long keys, keys that
    # differ only in the last inline word / byte
    keys = [(b"k" * L) for L in range(1, 301)] + [b"k" * 100, b"k" * 299, b"q" * 63, b"q" * 62,
                                                 b"q" * 45, b"q" * 46, b"q" * 47, b"q" * 48, b"q" * 45 + b"r", b"q" * 40 + b"r" + b"q" * 5]
    b = HostBatch(keys, 1, 3, 10_000, now)
    support.assert_results_equal(e.eval(b), o.eval(b), "ragged")
    support.assert_results_equal(e.eval(b), o.eval(b), "ragged again")
    # keys that differ only in the last byte / only in length
    keys = [b"same_prefix_" + bytes([65 + i]) for i in range(26)] + [b"same_prefix_", b"same_prefix_A\0"]
    b = HostBatch(keys * 3, 1, 2, 10_000, now)
    support.assert_results_equal(e.eval(b), o.eval(b), "near keys")
    # empty key and over-long key are per-item errors, everything else in the batch is unaffected
    b = HostBatch([b"ok_1", b"", b"z" * 301, b"ok_1"], 1, 5, 1000, now)
    r = e.eval(b)
    assert r.err[:4].tolist() == [0, 4, 7, 0] and r.remaining[:4].tolist() == [4, 0, 0, 3]
    assert e.size() == o.size() + 1
    e.close()


@pytest.mark.parametrize("flags", [1, 3, 5, 65])
def test_hash_collisions_are_resolved_exactly(flags):
    """GUBER_FLAG_TEST_WEAK_HASH keeps 6 bits of the key hash: hundreds of distinct keys share a
    tag, so the exact key verification, probing past a collision and the in-batch retry path all run."""
    o, e = Oracle(cache_size=1 << 16), engine(cache_size=4096, max_batch=4096, flags=flags)
    rng = np.random.default_rng(5)
    now = streams.NOW0
    for step in range(6):
        ids = rng.integers(0, 400, 2500)
        keys = [f"coll_{int(i)}" for i in ids]
        b = HostBatch(keys, 1, 7, 2_000, now + step * 900, algorithm=(ids % 2).astype(np.uint8))
        support.assert_results_equal(e.eval(b), o.eval(b), f"collisions step {step}")
    assert e.size() == o.size()
    assert e.stats()["retries"] > 0
    e.close()


def test_cache_operations_add_get_remove_each():
    # workers.go:537-626 AddCacheItem / GetCacheItem, lrucache.go:76-171
    o, e = Oracle(cache_size=1 << 16), engine(cache_size=4096, max_batch=1024)
    now = streams.NOW0
    items = [support.make_item(f"it_{i}", i % 2, limit=10 + i, duration=1000, remaining=5, remaining_f=2.5,
                               stamp=now - 10, burst=10 + i, expire_at=now + 1000 * (i % 3), status=i % 2)
             for i in range(50)]
    items.append(support.make_item("it_3", 0, limit=99, duration=7, remaining=1, stamp=now, expire_at=now + 5))
    items.append(support.make_item("nilval", 9, limit=1, expire_at=now + 50))   # unknown algorithm: nil Value
    want_ex = [o.add_item(it, now) for it in items]
    assert e.add_items(items) == want_ex
    assert e.size() == o.size() == 51
    for k in ["it_0", "it_3", "it_7", "missing", "nilval"]:
        assert e.get_item(k, now + 1) == o.get_item(k, now + 1), k
    # expired on access -> removed (lrucache.go:115-119)
    assert e.get_item("it_0", now + 1) is None and o.get_item("it_0", now + 1) is None
    assert e.size() == o.size()
    e.remove_item("it_7"); o.remove_item("it_7")
    assert e.get_item("it_7", now) is None and e.size() == o.size()
    key = lambda d: d["key"]
    assert sorted(e.each(), key=key) == sorted(o.each(), key=key)
    # evaluation continues from the injected state (UpdatePeerGlobals path, gubernator.go:425-459)
    b = HostBatch(["it_1", "it_2", "it_3", "nilval", "it_4"], 1, [11, 12, 99, 1, 14], 1000, now, algorithm=[1, 0, 0, 0, 0])
    got, want = e.eval(b), o.eval(b)
    support.assert_results_equal(got, want, "after add")
    assert got.counters() == want.counters()
    e.close()


@pytest.mark.parametrize("flags", [0, 2, 64])
def test_zipf_bench_stream_midsize(flags):
    """The BASELINE stream shape (Zipf 1.1, hits 1, limit 100, 60 s) at 200k keys / 16384 batch."""
    tab = streams.key_table(200_000)
    for algo in (0, 1):
        z = streams.ZipfSampler(200_000)
        o, e = Oracle(cache_size=1 << 21), engine(cache_size=400_000, max_batch=16384, flags=flags)
        for bi in range(10):
            b = streams.bench_batch(tab, z.draw(16384), streams.NOW0 + bi * 9_000, algorithm=algo)
            got, want = e.eval(b), o.eval(b)
            support.assert_results_equal(got, want, f"algo {algo} batch {bi}")
            assert got.counters() == want.counters()
        e.close()


def _routed_eval(torch, engines, sown, hb, ids, dev):
    """one host batch through S engines by ONE dispatcher call (guber_eval_batches_routed_dev, fused launches): split by the
    placement, per-shard device arrays, answers scattered back to the batch's order"""
    import ctypes as C
    S, n = len(engines), hb.n
    sh = sown[ids]
    L = int(hb.key_off[1] - hb.key_off[0])
    keymat = hb.key_bytes[:n * L].reshape(n, L)
    keep, batches, results, which, outs = [], [], [], [], []
    for j in range(S):
        pos = np.nonzero(sh == j)[0]
        if len(pos) == 0:
            continue
        m = len(pos)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        kb = t(np.concatenate([keymat[pos].reshape(-1), np.zeros(8, np.uint8)]))
        ko = t((np.arange(m + 1, dtype=np.int64) * L).astype(np.int32))
        cols = [t(hb.hits[pos]), t(hb.limit[pos]), t(hb.duration[pos]), t(hb.algorithm[pos]), t(hb.behavior[pos].view(np.int32))]
        out = [torch.empty(m, dtype=torch.uint8, device=dev), torch.empty(m, dtype=torch.int64, device=dev), torch.empty(m, dtype=torch.int64, device=dev),
               torch.empty(m, dtype=torch.int64, device=dev), torch.empty(m, dtype=torch.uint8, device=dev)]
        keep += [kb, ko] + cols + out
        batches.append(ga.GuberBatch(m, 0, kb.data_ptr(), ko.data_ptr(), cols[0].data_ptr(), cols[1].data_ptr(), cols[2].data_ptr(), None, None,
                                     cols[3].data_ptr(), cols[4].data_ptr(), None, None, None, hb.now_ms))
        results.append(ga.GuberResult(out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), out[4].data_ptr(), 0, 0, 0, 0, 0))
        which.append(j); outs.append((pos, out))
    torch.cuda.synchronize(dev)
    ga.Engine.eval_routed_dev(engines, (C.c_uint32 * len(which))(*which), (ga.GuberBatch * len(which))(*batches), (ga.GuberResult * len(which))(*results), len(which))
    for e in engines:
        e.synchronize()
    got = ga.HostResult(n)
    for pos, out in outs:
        for name, tns in zip(("status", "limit", "remaining", "reset_time", "err"), out):
            getattr(got, name)[pos] = tns.cpu().numpy()
    return got


def test_full_size_10m_keys_batch_65536():
    """BASELINE config 2/3 at full size ACROSS TIME: 10M resident keys, Zipf-1.1 batches of 65536, 16 batches per algorithm with
    the clock stepping 700 ms (leaky tokens leak: algorithms.go:356-371), one jump past `duration` (mass expiry and renewal:
    :106-147), a limit change and a duration change mid-run (:90-105, :296-320) — element-wise against the oracle, with the
    event counters, through ONE engine and through 12 engines behind one dispatcher (fused launches, the product's placement)."""
    import torch
    K, B, S = 10_000_000, 65536, 12
    dev = torch.device("cuda", 0)
    tab = streams.key_table(K)
    o, e = Oracle(cache_size=2 * K), engine(cache_size=K, max_batch=B)
    place = ga.Placement(S)
    place.observe_keys(*streams.keys_for_ids(tab, streams.ZipfSampler(K, seed=991).draw(1 << 20)))
    place.rebalance(0.125, True)
    sown = np.empty(K, np.uint8)
    for lo in range(0, K, 2_000_000):
        sown[lo:lo + 2_000_000] = place.route_keys(*streams.keys_for_ids(tab, np.arange(lo, min(K, lo + 2_000_000))))[0]
    per = np.bincount(sown, minlength=S)
    tstreams = [torch.cuda.Stream(device=dev) for _ in range(3)]        # shards that share a stream share their launches
    shards = [engine(cache_size=int(per[j]) + int(per[j]) // 4 + 1024, max_batch=B, stream=tstreams[j * 3 // S].cuda_stream) for j in range(S)]
    now = streams.NOW0
    # residency: one insert pass (hits 0 creates every bucket without consuming)
    for lo in range(0, K, B):
        ids = np.arange(lo, min(lo + B, K))
        b = streams.bench_batch(tab, ids, now, hits=0)
        o.eval(b); e.eval(b)
        _routed_eval(torch, shards, sown, b, ids, dev)
    assert e.size() == o.size() == K == sum(x.size() for x in shards)
    z = streams.ZipfSampler(K)
    for algo in (0, 1):
        t = now
        c0 = (e.counters()[:3], [x.counters()[:3] for x in shards], o.counters()[:3])
        for bi in range(16):
            t += 700 if bi != 9 else 61_000                          # batch 9: every bucket of this algorithm's earlier batches has expired
            limit = 100 if bi < 5 else 40 if bi < 12 else 250        # limit changes mid-run (down, then up)
            duration = 60_000 if bi < 7 else 30_000                  # duration change mid-run
            ids = z.draw(B)
            b = streams.bench_batch(tab, ids, t, algorithm=algo, limit=limit, duration=duration)
            want = o.eval(b)
            support.assert_results_equal(e.eval(b), want, f"one engine, algo {algo} batch {bi}")
            support.assert_results_equal(_routed_eval(torch, shards, sown, b, ids, dev), want, f"12 engines, algo {algo} batch {bi}")
            if algo == 0 and bi < 5:
                # per key: remaining strictly decreases by 1 per admitted request, in request order
                order = np.argsort(ids, kind="stable")
                sid, rem, st = ids[order], want.remaining[:B][order], want.status[:B][order]
                same = sid[1:] == sid[:-1]
                under = (st[1:] == 0) & (st[:-1] == 0) & same
                assert (rem[:-1][under] - rem[1:][under] == 1).all()
        d_one = tuple(a - b_ for a, b_ in zip(e.counters()[:3], c0[0]))
        d_many = tuple(sum(x.counters()[:3][k] - c[k] for x, c in zip(shards, c0[1])) for k in range(3))
        d_orc = tuple(a - b_ for a, b_ in zip(o.counters()[:3], c0[2]))
        assert d_one == d_many == d_orc, (algo, d_one, d_many, d_orc)   # over-limit / cache hit / cache miss events
        now = t + 120_000
    assert e.size() == o.size() == sum(x.size() for x in shards)
    assert sum(x.stats()["retries"] for x in shards) == 0 and sum(x.stats()["fused_batches"] for x in shards) > 0
    for x in shards:
        x.close()
    place.close()
    e.close()


def test_large_batch_uses_radix_pipeline():
    """Batches above 65536 requests take the global radix-sort kernel sequence (3 digit passes)."""
    K, B = 300_000, 200_000
    tab = streams.key_table(K)
    z = streams.ZipfSampler(K, seed=7)
    o, e = Oracle(cache_size=1 << 21), engine(cache_size=2 * K, max_batch=B)
    for bi in range(3):
        b = streams.bench_batch(tab, z.draw(B), streams.NOW0 + bi * 25_000, algorithm=bi % 2)
        got, want = e.eval(b), o.eval(b)
        support.assert_results_equal(got, want, f"large batch {bi}")
        assert got.counters() == want.counters()
    e.close()


def test_device_router_matches_host_ring():
    """k_route (ring staged in LDS, fnv1/fnv1a + binary search) == guber_ring_route == the reference's
    ReplicatedConsistentHash.Get (replicated_hash.go:104-119), incl. the known-answer distribution."""
    import torch
    k = scenarios.load("kat_vectors.json")["ring_distribution"]
    e = engine(cache_size=1024, max_batch=1024)
    keys = [f"192.168.{(i >> 8) & 255}.{i & 255}" for i in range(k["n_keys"])]
    hb = HostBatch(keys, 0, 0, 0, 0)
    dev = torch.device("cuda", 0)
    d_kb = torch.from_numpy(hb.key_bytes).to(dev)
    d_ko = torch.from_numpy(hb.key_off.view(np.int32)).to(dev)
    for kind in ("fnv1", "fnv1a"):
        ring = ga.Ring(k["hosts"], k["replicas"], kind)
        d_owner = torch.empty(len(keys), dtype=torch.int32, device=dev)
        e.route_dev(ring, d_kb.data_ptr(), d_ko.data_ptr(), len(keys), d_owner.data_ptr())
        owner = d_owner.cpu().numpy()
        assert np.array_equal(owner, ring.route(keys).astype(np.int32))
        assert {h: int((owner == i).sum()) for i, h in enumerate(k["hosts"])} == k[kind]
    ring8 = ga.Ring([f"gpu{i}" for i in range(8)], 512, "fnv1")
    tab = streams.key_table(100_000)
    kb, ko = streams.keys_for_ids(tab, np.arange(100_000))
    d_kb, d_ko = torch.from_numpy(kb).to(dev), torch.from_numpy(ko.view(np.int32)).to(dev)
    d_owner = torch.empty(100_000, dtype=torch.int32, device=dev)
    e.route_dev(ring8, d_kb.data_ptr(), d_ko.data_ptr(), 100_000, d_owner.data_ptr())
    assert np.array_equal(d_owner.cpu().numpy(), ring8.route((kb, ko)).astype(np.int32))
    e.close()


def test_global_behaviour_engines_vs_model():
    """BASELINE config 5 semantics: N logical GPUs (N engines on this device) with device-side GLOBAL
    queues (guber_global_take) and the product orchestrator, against the global.go model; also the
    GLOBAL vectors of the reference's functional tests."""
    import test_global as tg
    from global_model import GlobalModel
    import pyglobal as global_sync
    n = 4
    ring = ga.Ring([f"gpu{i}" for i in range(6)])
    mk = lambda: engine(cache_size=4096, max_batch=4096, max_key_bytes=64, flags=ga.FLAG_GLOBAL)
    cluster = global_sync.LocalCluster([mk() for _ in range(6)], ring)
    assert tg.run_vectors(lambda r, q, now: tg.cluster_request(cluster, r, q, now), cluster.sync, ring, 6) >= 40
    for seed in (1, 2, 3):
        ring4 = ga.Ring([f"gpu{i}" for i in range(n)])
        cl = global_sync.LocalCluster([mk() for _ in range(n)], ring4)
        model = GlobalModel(n, lambda k: int(ring4.route([k])[0]))
        tg.run_random(lambda r, b, now: cl.ranks[r].evaluate(b["keys"], b["hits"], b["limit"], b["duration"], now,
                                                             algorithm=b["algorithm"], behavior=b["behavior"], burst=0,
                                                             created_at=now),
                      cl.sync, model, n, seed, steps=200)
        # replicas converge: after a final sync every peer reports the same remaining for every key
        cl.sync(tg.NOW + 10_000); model.sync(tg.NOW + 10_000)
        for k in range(40):
            key = f"glob_{k}".encode()
            vals = {r: (cl.ranks[r].node.get_item(key, tg.NOW + 10_000) or {}).get("remaining") for r in range(n)}
            want = {r: (model.oracles[r].get_item(key, tg.NOW + 10_000) or {}).get("remaining") for r in range(n)}
            assert vals == want, (seed, key, vals, want)


@pytest.mark.parametrize("flags", [0, 2, 64])
def test_created_at_only_variation_on_hot_keys(flags):
    rng = np.random.default_rng(17)
    now = streams.NOW0
    for algo in (0, 1):
        o, e = Oracle(cache_size=1 << 12), engine(cache_size=1024, max_batch=4096, flags=flags)
        for step in range(6):
            n = 3000
            keys = [b"hotk"] * 2500 + [f"cold{i}".encode() for i in range(500)]
            created = now + step * 400 + np.sort(rng.integers(0, 3, n))
            beh = 8 if step == 4 else 0
            dur = 60_000 if step != 3 else 30_000
            b = HostBatch(keys, 1, 5000, dur, now + step * 400 + 2, created_at=created, algorithm=algo, behavior=beh)
            support.assert_results_equal(e.eval(b), o.eval(b), f"algo {algo} step {step}")
        e.close()


def test_config0_batch_of_one_1k_keys():
    """BASELINE configs[0] (benchmark_test.go:63-84 shape): TOKEN_BUCKET, batch = 1, limit 10, 5 s, hits 1,
    1000 keys cycled through the C ABI one request per call."""
    o, e = Oracle(cache_size=1 << 12), engine(cache_size=4096, max_batch=1024)
    now = streams.NOW0
    for i in range(3000):
        b = HostBatch([f"bench_{i % 1000:04d}"], 1, 10, 5000, now + i)
        assert e.eval(b).rows() == o.eval(b).rows(), i
    e.close()


def test_back_to_back_device_batches_match_oracle():
    """Many dependent device-resident batches (guber_eval_batch_dev) enqueued back to back without any host
    synchronisation in between, on overlapping key sets, checked batch by batch against the oracle."""
    import torch
    dev = torch.device("cuda", 0)
    K, B, steps = 3000, 4096, 40
    tab = streams.key_table(K)
    z = streams.ZipfSampler(K, seed=5)
    stream = torch.cuda.Stream(device=dev)
    e = ga.Engine(cache_size=4 * K, max_batch=B, stream=stream.cuda_stream)
    o = Oracle(cache_size=1 << 16)
    hbs, dbs, outs = [], [], []
    for s in range(steps):
        hb = streams.bench_batch(tab, z.draw(B), streams.NOW0 + s * 900, algorithm=s % 2, limit=40, duration=5000)
        t = [torch.from_numpy(hb.key_bytes).to(dev), torch.from_numpy(hb.key_off.view(np.int32)).to(dev),
             torch.from_numpy(hb.hits).to(dev), torch.from_numpy(hb.limit).to(dev), torch.from_numpy(hb.duration).to(dev),
             torch.from_numpy(hb.algorithm).to(dev), torch.from_numpy(hb.behavior.view(np.int32)).to(dev)]
        p = [x.data_ptr() for x in t]
        r = dict(status=torch.empty(B, dtype=torch.uint8, device=dev), err=torch.empty(B, dtype=torch.uint8, device=dev),
                 limit=torch.empty(B, dtype=torch.int64, device=dev), remaining=torch.empty(B, dtype=torch.int64, device=dev),
                 reset_time=torch.empty(B, dtype=torch.int64, device=dev))
        hbs.append(hb); dbs.append((t, ga.GuberBatch(B, 0, p[0], p[1], p[2], p[3], p[4], None, None, p[5], p[6], None, None, None, hb.now_ms)))
        outs.append((r, ga.GuberResult(r["status"].data_ptr(), r["limit"].data_ptr(), r["remaining"].data_ptr(),
                                       r["reset_time"].data_ptr(), r["err"].data_ptr(), 0, 0, 0, 0, 0)))
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(stream):
        for s in range(steps // 2):
            e.eval_dev(dbs[s][1], outs[s][1])
        # the rest as ONE queue of batches (guber_eval_batches_dev)
        rest = list(range(steps // 2, steps))
        ba = (ga.GuberBatch * len(rest))(*[dbs[s][1] for s in rest])
        ra = (ga.GuberResult * len(rest))(*[outs[s][1] for s in rest])
        e.eval_many_dev(ba, ra, len(rest))
    e.synchronize()
    for s in range(steps):
        want = o.eval(hbs[s])
        got = ga.HostResult(B)
        for name in ("status", "limit", "remaining", "reset_time", "err"):
            getattr(got, name)[:] = outs[s][0][name].cpu().numpy()
        support.assert_results_equal(got, want, f"step {s}")
    assert e.size() == o.size()
    e.close()


@pytest.mark.parametrize("n_engines,shared_stream", [(3, True), (6, True), (4, False)])
def test_routed_batches_of_several_engines_share_launches(n_engines, shared_stream):
    """guber_eval_batches_routed_dev: one dispatcher, several engines (logical shards with disjoint keys).  Engines on one
    stream get up to four batches per pair of launches (k_front_multi / k_eval2_multi); every batch must equal the oracle
    of ITS engine evaluated in the engine's own order — hot keys, both algorithms, ragged batch sizes, a batch too small
    and one too large for the two-launch pipeline in between."""
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    K, B, steps = 2500, 4096, 36
    rng = np.random.default_rng(11)
    tab = streams.key_table(K * n_engines)
    stream = torch.cuda.Stream(device=dev)
    strs = [stream if shared_stream else torch.cuda.Stream(device=dev) for _ in range(n_engines)]
    engs = [ga.Engine(cache_size=1 << 18, max_batch=3 * B, stream=strs[j].cuda_stream) for j in range(n_engines)]     # (room for every request to be a new key: no batch waits for an eviction pre-pass)
    orcs = [Oracle(cache_size=1 << 16) for _ in range(n_engines)]
    zs = [streams.ZipfSampler(K, seed=100 + j) for j in range(n_engines)]
    which, hbs, keep, cb, cr = [], [], [], [], []
    for s in range(steps):
        j = int(rng.integers(0, n_engines)) if s % 5 else s % n_engines
        n = [B, B, 1000, B, 3 * B, 200, B][s % 7]                       # 200: one-tile batch; 3 * B: still two launches (<= 65 536)
        hb = streams.bench_batch(tab, j * K + zs[j].draw(n), streams.NOW0 + s * 700, algorithm=s % 2, limit=30, duration=4000)
        t = [torch.from_numpy(hb.key_bytes).to(dev), torch.from_numpy(hb.key_off.view(np.int32)).to(dev),
             torch.from_numpy(hb.hits).to(dev), torch.from_numpy(hb.limit).to(dev), torch.from_numpy(hb.duration).to(dev),
             torch.from_numpy(hb.algorithm).to(dev), torch.from_numpy(hb.behavior.view(np.int32)).to(dev)]
        p = [x.data_ptr() for x in t]
        r = dict(status=torch.empty(n, dtype=torch.uint8, device=dev), err=torch.empty(n, dtype=torch.uint8, device=dev),
                 limit=torch.empty(n, dtype=torch.int64, device=dev), remaining=torch.empty(n, dtype=torch.int64, device=dev),
                 reset_time=torch.empty(n, dtype=torch.int64, device=dev))
        keep.append((t, r)); which.append(j); hbs.append(hb)
        cb.append(ga.GuberBatch(n, 0, p[0], p[1], p[2], p[3], p[4], None, None, p[5], p[6], None, None, None, hb.now_ms))
        cr.append(ga.GuberResult(r["status"].data_ptr(), r["limit"].data_ptr(), r["remaining"].data_ptr(),
                                 r["reset_time"].data_ptr(), r["err"].data_ptr(), 0, 0, 0, 0, 0))
    torch.cuda.synchronize(dev)
    wa = (C.c_uint32 * steps)(*which)
    ba = (ga.GuberBatch * steps)(*cb)
    ra = (ga.GuberResult * steps)(*cr)
    ga.Engine.eval_routed_dev(engs, wa, ba, ra, steps)
    for e in engs:
        e.synchronize()
    sums = [[0, 0, 0] for _ in range(n_engines)]
    for s in range(steps):
        want = orcs[which[s]].eval(hbs[s])
        got = ga.HostResult(hbs[s].n)
        for name in ("status", "limit", "remaining", "reset_time", "err"):
            getattr(got, name)[:] = keep[s][1][name].cpu().numpy()
        support.assert_results_equal(got, want, f"batch {s} of engine {which[s]}")
        for q in range(3):
            sums[which[s]][q] += want.counters()[q]
    fused = sum(e.stats()["fused_batches"] for e in engs)
    assert (fused > steps // 3) if shared_stream else fused == 0, fused
    for j, (e, o) in enumerate(zip(engs, orcs)):
        assert e.size() == o.size()
        assert list(e.counters()[:3]) == sums[j], (j, e.counters(), sums[j])
        e.close()


def test_compaction_keeps_churning_key_sets_going():
    """A key population that keeps changing: the directory would fill with expired / removed buckets; the
    table rebuilds itself (guber_compact, also automatic) and results stay identical to the oracle."""
    o, e = Oracle(cache_size=2048), engine(cache_size=2048, max_batch=1024, table_slots=4096)     # 4096 slots, limit 3584 tags
    now = streams.NOW0
    for step in range(40):
        keys = [f"churn_{step}_{i}" for i in range(400)] + [f"steady_{i}" for i in range(100)]
        b = HostBatch(keys, 1, 5, [20] * 400 + [600_000] * 100, now, algorithm=[step % 2] * 400 + [0] * 100)
        support.assert_results_equal(e.eval(b), o.eval(b), f"churn step {step}")
        now += 1000
    st = e.stats()
    assert st["compactions"] >= 3 and st["tags_used"] <= 3584
    e.compact(now)                                        # explicit: only the items of the list keep an entry (expired ones are still items: lrucache.go:76-85, :159)
    assert e.size() == o.size() == 2048 and e.stats()["tags_used"] == 2048
    b = HostBatch([f"steady_{i}" for i in range(100)], 1, 5, 600_000, now)
    support.assert_results_equal(e.eval(b), o.eval(b), "after compaction")
    assert sorted(d["key"] for d in e.each()) == sorted(d["key"] for d in o.each())
    e.close()


def test_store_events_golden_teststore():
    """store_test.go TestStore on the engine: guber_probe_missing -> Store.Get -> guber_add_items ->
    guber_eval_batch_store -> Remove / OnChange, the call sequence and the OnChange item of the reference's mock."""
    assert scenarios.run_store_events(lambda: engine(cache_size=4096, max_batch=1024)) == 10


@pytest.mark.parametrize("flags", [0, 2, 64])
def test_store_events_match_the_oracle_on_random_batches(flags):
    """Every request's Store callbacks (which, in which order, and the CacheItem handed to OnChange — the state right
    after THAT request, also in the middle of a run on a hot key) equal the reference restatement's."""
    rng = np.random.default_rng(77 + flags)
    o, e = Oracle(cache_size=1 << 16), engine(cache_size=1 << 14, max_batch=8192, flags=flags)
    so, se = support.MockStore(write_through=True), support.MockStore(write_through=True)
    now = streams.NOW0
    for step in range(12):
        n = int(rng.integers(1, 3000))
        kid = rng.zipf(1.3, n) % 400
        # per key one request shape per batch half, so that hot keys form long uniform runs as well as mixed ones
        shape = rng.integers(0, 6, 400)
        mixed = rng.random(n) < 0.15
        algo = np.where(mixed, rng.integers(0, 2, n), shape[kid] & 1).astype(np.uint8)
        beh = np.where(mixed, rng.choice([0, 8, 32, 0, 0], n), np.where(shape[kid] == 5, 32, 0)).astype(np.uint32)
        hits = np.where(mixed, rng.integers(0, 4, n), 1).astype(np.int64)
        owner = np.where(rng.random(n) < 0.1, 0, 1).astype(np.uint8) if step % 3 == 2 else np.ones(n, np.uint8)
        keys = [b"st_k%d" % k for k in kid]
        b = HostBatch(keys, hits, 20 + (kid % 7), np.where(kid % 11 == 0, 3, 60_000), now, algorithm=algo, behavior=beh, is_owner=owner,
                      burst=np.zeros(n, np.int64), created_at=np.full(n, now))
        for st in (so, se):
            st.calls.clear()
            st.now = now
        ro, re_ = o.eval_store(b, so), e.eval_store(b, se)
        support.assert_results_equal(re_, ro, f"store batch {step}")
        pick = lambda st, kinds: [c for c in st.calls if c[0] in kinds]
        assert pick(se, ("on_change", "remove")) == pick(so, ("on_change", "remove")), f"batch {step}"
        # Store.Get: the engine asks once per non-resident key, before the batch; the reference asks again when a key is
        # requested after an in-batch token RESET_REMAINING dropped it (documented divergence, the answer cannot differ
        # for a store that honours Remove)
        first_req = {}
        for i, k in enumerate(keys):
            first_req.setdefault(k.decode(), i)
        want = [c for c in pick(so, ("get",)) if first_req[c[2]] == c[1]]
        assert sorted(pick(se, ("get",))) == sorted(want), f"batch {step}"
        now += int(rng.choice([1, 2, 5, 4000]))
    e.close()


@pytest.mark.parametrize("flags", [0, 2, 64])
def test_created_at_only_variation_on_hot_leaky_keys(flags):
    """Hot LEAKY keys whose requests carry slightly different created_at (aggregated RPC payloads): parallel path
    while no request leaks (leaky_created_harmless), serial otherwise; bit-exact either way."""
    rng = np.random.default_rng(23 + flags)
    o, e = Oracle(cache_size=1 << 16), engine(cache_size=1 << 14, max_batch=16384, flags=flags)
    t = streams.NOW0
    for step in range(16):
        n = int(rng.integers(2000, 12000))
        kid = rng.zipf(1.2, n) % 50
        dur = np.where(kid % 5 == 0, 3, 60_000)
        kind = step % 8
        slice_of = np.arange(n) // 1000                                      # one RPC payload = 1000 items
        created = t - 2 + slice_of % 4                                       # harmless: RPCs stamped 0..3 ms apart
        if kind == 2:
            created = created + np.where(rng.random(n) < 0.001, 700, 0)      # a few members leak
        elif kind == 3:
            created = np.where(rng.random(n) < 0.001, t - 700_000, created)  # far in the past
        elif kind == 4:
            created = t + rng.integers(-2000, 2000, n)
        elif kind == 5:
            created[np.unique(kid, return_index=True)[1]] += 200_000         # the first toucher of every key is the odd one
        hits = np.where(kind == 6, 0, 1)
        b = HostBatch([b"hl_%d" % k for k in kid], hits, 100 + kid % 3, dur, t, created_at=created,
                      algorithm=np.where(kid % 7 == 3, 0, 1).astype(np.uint8), behavior=np.where(step == 9, 32, 0).astype(np.uint32))
        support.assert_results_equal(e.eval(b), o.eval(b), f"step {step} kind {kind}")
        assert e.size() == o.size()
        t += int(rng.choice([1, 3, 50, 30_001, 120_000]))
    e.close()


def test_global_behaviour_device_resident_exchange_vs_model():
    """The same GLOBAL semantics with the exchange running on device arrays (guber_global_take_dev ->
    guber_ring_route_rows_dev -> row exchange -> guber_eval_batch_dev -> guber_add_items_dev): logical ranks on one
    device against the global.go model, the reference's GLOBAL vectors, and a 1-rank RCCL process group."""
    import torch
    import test_global as tg
    from global_model import GlobalModel
    import pyglobal_dev as gsd
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        mk = lambda: ga.Engine(cache_size=4096, max_batch=4096, max_key_bytes=64, flags=ga.FLAG_GLOBAL, stream=s.cuda_stream)
        ring = ga.Ring([f"gpu{i}" for i in range(6)])
        cluster = gsd.LocalClusterDev([mk() for _ in range(6)], ring, "cuda:0")
        assert tg.run_vectors(lambda r, q, now: tg.cluster_request(cluster, r, q, now), cluster.sync, ring, 6) >= 40
        n = 4
        for seed in (1, 2):
            ring4 = ga.Ring([f"gpu{i}" for i in range(n)])
            cl = gsd.LocalClusterDev([mk() for _ in range(n)], ring4, "cuda:0")
            model = GlobalModel(n, lambda k: int(ring4.route([k])[0]))
            tg.run_random(lambda r, b, now: cl.ranks[r].evaluate(b["keys"], b["hits"], b["limit"], b["duration"], now,
                                                                 algorithm=b["algorithm"], behavior=b["behavior"], burst=0,
                                                                 created_at=now),
                          cl.sync, model, n, seed, steps=200)
            cl.sync(tg.NOW + 10_000); model.sync(tg.NOW + 10_000)
            for k in range(40):
                key = f"glob_{k}".encode()
                vals = {r: (cl.ranks[r].node.get_item(key, tg.NOW + 10_000) or {}).get("remaining") for r in range(n)}
                want = {r: (model.oracles[r].get_item(key, tg.NOW + 10_000) or {}).get("remaining") for r in range(n)}
                assert vals == want, (seed, key, vals, want)
            assert sum(r.fallbacks for r in cl.ranks) == 0
        # the collectives themselves: one rank, RCCL backend (all_to_all_single / all_gather_into_tensor on device tensors)
        import torch.distributed as dist
        import socket
        sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        try:
            ring1 = ga.Ring(["gpu0"])
            g = gsd.GlobalSyncDev(mk(), 0, 1, ring1, gsd.TorchTransportDev("cuda:0"), "cuda:0")
            model = GlobalModel(1, lambda k: 0)
            tg.run_random(lambda r, b, now: g.evaluate(b["keys"], b["hits"], b["limit"], b["duration"], now, algorithm=b["algorithm"],
                                                       behavior=b["behavior"], burst=0, created_at=now),
                          lambda now: [g.sync(now)], model, 1, 5, steps=60)
        finally:
            dist.destroy_process_group()


def test_global_behaviour_native_exchange_vs_model():
    """guber_global_sync (C++: device kernels + one exchange, include/guber_gpu.h) with N logical ranks of this process on one
    GPU (device-copy transport; RCCL needs distinct GPUs): the reference's GLOBAL vectors — who sends hits, who broadcasts —
    and random streams against the global.go model, replicas converging, no host fallbacks; then 200 000 pending rows
    through one sync in under a millisecond per ... (timing is reported, the bound checked loosely)."""
    import time
    import test_global as tg
    from global_model import GlobalModel
    from gubernator_amd import global_native as gn
    mk = lambda: engine(cache_size=4096, max_batch=4096, max_key_bytes=64, flags=ga.FLAG_GLOBAL)
    ring = ga.Ring([f"gpu{i}" for i in range(6)])
    cluster = gn.Comm.local([mk() for _ in range(6)], ring)
    assert tg.run_vectors(lambda r, q, now: tg.cluster_request(cluster, r, q, now), cluster.sync, ring, 6) >= 40
    cluster.close()
    n = 4
    for seed in (1, 2, 3):
        ring4 = ga.Ring([f"gpu{i}" for i in range(n)])
        cl = gn.Comm.local([mk() for _ in range(n)], ring4)
        model = GlobalModel(n, lambda k: int(ring4.route([k])[0]))
        tg.run_random(lambda r, b, now: cl.ranks[r].evaluate(b["keys"], b["hits"], b["limit"], b["duration"], now,
                                                             algorithm=b["algorithm"], behavior=b["behavior"], burst=0,
                                                             created_at=now),
                      cl.sync, model, n, seed, steps=200)
        cl.sync(tg.NOW + 10_000); model.sync(tg.NOW + 10_000)
        for k in range(40):
            key = f"glob_{k}".encode()
            vals = {r: (cl.ranks[r].node.get_item(key, tg.NOW + 10_000) or {}).get("remaining") for r in range(n)}
            want = {r: (model.oracles[r].get_item(key, tg.NOW + 10_000) or {}).get("remaining") for r in range(n)}
            assert vals == want, (seed, key, vals, want)
        assert cl.last["fallbacks"] == 0
        cl.close()
    # volume: 2 ranks, 100 000 distinct GLOBAL keys hit on each rank -> ~200 000 rows per sync
    K = 100_000
    ring2 = ga.Ring(["gpu0", "gpu1"])
    big = lambda: engine(cache_size=2 * K, max_batch=65536, max_key_bytes=64, flags=ga.FLAG_GLOBAL)
    cl = gn.Comm.local([big(), big()], ring2)
    tab = streams.key_table(K)
    now = streams.NOW0
    ms = []
    for rnd in range(4):
        for r in range(2):
            for lo in range(0, K, 65536):
                ids = np.arange(lo, min(K, lo + 65536))
                kb, ko = streams.keys_for_ids(tab, ids)
                cl.ranks[r].evaluate((kb, ko), 1, 1000, 600_000, now)
        t0 = time.perf_counter()
        st = cl.sync(now)
        ms.append((time.perf_counter() - t0) * 1e3)
        assert sum(s_["hits_sent"] for s_ in st) == K and sum(s_["hits_applied"] for s_ in st) == K, st   # every key: one row from its non-owner
        assert sum(s_["broadcast"] for s_ in st) == K and sum(s_["installed"] for s_ in st) == K, st
        now += 10
    for r in range(2):       # both replicas agree with what 2 hits per key and round leave
        it = cl.ranks[r].node.get_item(bytes(tab[12345]), now)
        assert it["remaining"] == 1000 - 2 * 4, it
    print("native global sync, 2 ranks, 100k rows each way:", [round(x, 3) for x in ms], "ms")
    assert min(ms) < 5.0, ms
    # a tick holds every local engine's lock, taken in address order — the order fused launches over the same engines use: a
    # dispatcher hammering guber_eval_batches_routed_dev on both engines while ticks run must not deadlock
    import threading
    stop = threading.Event()
    engines = [r_.node for r_ in cl.ranks]
    kb, ko = streams.keys_for_ids(tab, np.arange(4096))
    def hammer():
        while not stop.is_set():
            for e_ in reversed(engines):
                e_.eval(ga.HostBatch((kb, ko), 0, 1000, 600_000, now))
    th = threading.Thread(target=hammer)
    th.start()
    for _ in range(20):
        cl.sync(now)
    stop.set()
    th.join(timeout=30)
    assert not th.is_alive()
    cl.close()
    e1 = big()
    with pytest.raises(ga.GuberError):                                   # one engine cannot be two ranks
        gn.Comm.local([e1, e1], ring2)
    e1.close()


def test_gregorian_intervals_on_the_device():
    """DURATION_IS_GREGORIAN evaluated in the kernels from the batch clock (no greg_expire / greg_duration arrays): equal to the
    oracle fed with the host-computed calendar values — minutes .. years, the weeks / invalid-interval errors, both pipelines."""
    from test_kernel_logic_host import _gregorian_batches
    for flags in (0, 2, 64):
        o, e = Oracle(cache_size=1 << 12), engine(cache_size=1024, max_batch=1024, flags=flags)
        for bi, (with_vals, without) in enumerate(_gregorian_batches(6 + flags)):
            support.assert_results_equal(e.eval(without), o.eval(with_vals), f"flags {flags} batch {bi}")
        e.close()


def test_gregorian_intervals_in_the_daemons_zone_on_the_device():
    """guber_set_timezone: DURATION_IS_GREGORIAN intervals end where the DAEMON's zone says (interval.go uses now.Location()).  The
    kernels — one-launch path, two-launch and owner-partitioned pipelines — evaluate requests without precomputed calendar values in
    America/New_York from a clock that crosses the start of daylight saving time; the oracle is fed the values of an independent
    restatement of Go's time.Date (tests/test_timezone_cpu.py, itself checked against zoneinfo)."""
    import test_timezone_cpu as tz
    from test_kernel_logic_host import _gregorian_batches
    zone = ga.zone_transitions("America/New_York", 2018, 2021)
    def greg(now_ms, d):
        if d == 3:
            return 0, -2
        if d not in (0, 1, 2, 4, 5):
            return 0, -3
        return tz.go_expiration(zone, now_ms * 1_000_000, d), tz.go_duration(zone, now_ms * 1_000_000, d)
    ga.set_timezone(*zone)
    try:
        now0 = (zone[1][2][0] - 3 * 3600) * 1000                      # three hours before the spring-forward of 2019
        for flags in (0, 32, 64):
            o, e = Oracle(cache_size=1 << 12), engine(cache_size=1024, max_batch=1024, flags=flags)
            for bi, (with_vals, without) in enumerate(_gregorian_batches(21 + flags, n_batches=30, greg_fn=greg, now0=now0)):
                support.assert_results_equal(e.eval(without), o.eval(with_vals), f"flags {flags} batch {bi}")
            e.close()
    finally:
        ga.set_timezone()


def test_small_batches_one_launch_path():
    """Batches of <= 256 requests (BASELINE configs[0]: one request per call, benchmark_test.go:63-84) take one launch of one
    workgroup; duplicate-heavy, mixed-algorithm, error-carrying small batches equal the oracle, and what the path declines
    (requests of one key that differ) is answered by the general pipeline with the same result."""
    rng = np.random.default_rng(41)
    o, e = Oracle(cache_size=1 << 16), engine(cache_size=4096, max_batch=4096)
    now = streams.NOW0
    for step in range(400):
        n = int(rng.choice([1, 1, 2, 3, 17, 64, 255, 256]))
        ids = rng.integers(0, 30, n)
        keys = [b"small_%d" % int(i) for i in ids]
        uniform = rng.random() < 0.7
        hits = np.full(n, int(rng.choice([0, 1, 2, 5]))) if uniform else rng.choice([0, 1, 2, 5], n)
        algo = (ids % 2).astype(np.uint8) if rng.random() < 0.9 else rng.choice([0, 1, 7], n).astype(np.uint8)
        b = HostBatch(keys, hits, 20, int(rng.choice([50, 5000])), now, algorithm=algo, behavior=int(rng.choice([0, 0, 32])))
        got, want = e.eval(b), o.eval(b)
        support.assert_results_equal(got, want, f"step {step} n {n}")
        assert got.counters()[:3] == want.counters()[:3], step
        now += int(rng.choice([0, 1, 40, 6000]))
    assert e.size() == o.size()
    e.close()


@pytest.mark.parametrize("flags", [0, 1])
def test_stage_path_two_in_flight_matches_oracle(flags):
    """guber_stage_*: request arrays written in place into device-visible host memory, two batches in flight, responses read in
    place — against the oracle, batch sizes on both sides of the one-launch limit; with the weak test hash (flags 1) distinct
    keys collide and guber_stage_wait has to resolve the internal retries."""
    rng = np.random.default_rng(77 + flags)
    o, e = Oracle(cache_size=1 << 16), engine(cache_size=8192, max_batch=4096, flags=flags)
    stages = [ga.Stage(e, 4096, key_bytes_cap=4096 * 24) for _ in range(2)]
    now = streams.NOW0
    pending = None
    for step in range(120):
        n = int(rng.choice([1, 3, 200, 256, 257, 1000, 4096]))
        ids = rng.integers(0, 60 if flags else 400, n)
        keys = [b"stage_%d" % int(i) for i in ids]
        hits = np.full(n, int(rng.choice([0, 1, 2]))) if rng.random() < 0.8 else rng.choice([0, 1, 3], n)
        hb = HostBatch(keys, hits, 50, int(rng.choice([100, 60_000])), now, algorithm=(ids % 2).astype(np.uint8), burst=0,
                       created_at=now, is_owner=1, behavior=int(rng.choice([0, 32])))
        st = stages[step % 2]
        st.fill(hb)
        st.submit()
        if flags:                        # colliding keys go through the internal retry in wait(): strict order needs one stage in flight
            st.wait()
            support.assert_results_equal(st.result(), o.eval(hb), f"flags {flags} step {step} n {n}")
            now += int(rng.choice([0, 1, 30, 700]))
            continue
        if pending is not None:
            pst, phb, pstep = pending
            pst.wait()
            got, want = pst.result(), o.eval(phb)
            support.assert_results_equal(got, want, f"flags {flags} step {pstep} n {phb.n}")
            assert got.counters()[:3] == want.counters()[:3], (pstep, got.counters(), want.counters())
        pending = (st, hb, step)
        now += int(rng.choice([0, 1, 30, 700]))
    if pending is not None:
        pst, phb, pstep = pending
        pst.wait()
        support.assert_results_equal(pst.result(), o.eval(phb), f"flags {flags} last step")
    assert e.size() == o.size()
    for st in stages:
        st.close()
    e.close()


def test_claim_table_epoch_wraps():
    """The claim table's cells are tagged with a 16-bit batch epoch; after 65 535 batches the table is wiped and the
    epoch restarts.  66 500 small batches (duplicates inside each) straddle the wrap and must stay bit-exact."""
    o, e = Oracle(cache_size=1 << 12), engine(cache_size=1024, max_batch=1024)
    rng = np.random.default_rng(9)
    now = streams.NOW0
    keys = [b"wrap_%d" % i for i in range(40)]
    for i in range(66_500):
        n = 3 if i % 1000 else 200
        ks = [keys[j] for j in rng.integers(0, 40, n)]
        b = HostBatch(ks, 1, 1_000_000, 3_600_000, now + i // 10)
        if i % 1000 == 0 or i > 65_400:
            support.assert_results_equal(e.eval(b), o.eval(b), f"batch {i}")
        else:
            e.eval(b); o.eval(b)
    e.close()


def test_lrucache_vectors_on_the_engine():
    """lrucache_test.go TestLRUCache through guber_add_items / guber_get_item / guber_remove_item / guber_size, the two
    eviction cases included (lrucache_test.go:339-428): adding an 11th item to a cache of 10 evicts exactly one — the least
    recently used — and gubernator_unexpired_evictions_count moves only when the victim had not expired."""
    assert scenarios.run_cache_vectors(lambda cs: engine(cache_size=cs or 4096, max_batch=1024), evicting=True) > 3000


def test_live_set_larger_than_the_cache_is_served_by_evicting():
    """A key population that outgrows CacheSize: the reference evicts the least recently used items and keeps answering
    (lrucache.go:98-100); so does the engine — no GUBER_ITEM_E_TABLE_FULL, the size stays bounded, keys touched in every batch
    survive, evictions of live items are counted.  Every answer equals the oracle's (a bounded LRU of the same size)."""
    cs = 2000
    o, e = Oracle(cache_size=cs), engine(cache_size=cs, max_batch=1024)
    now = streams.NOW0
    for step in range(24):
        keys = [f"steady_{i}" for i in range(100)] + [f"churn_{step}_{i}" for i in range(900)]
        b = HostBatch(keys, 1, 50, 3_600_000, now)
        got, want = e.eval(b), o.eval(b)
        support.assert_results_equal(got, want, f"churn step {step}")
        assert (got.err[:b.n] == 0).all()
        st = e.stats()
        assert st["cache_size"] <= cs, (step, st)
        assert st["cache_size"] >= min(cs - cs // 16, 1000 * (step + 1)), (step, st)
        now += 1000
    st = e.stats()
    assert st["unexpired_evictions"] >= 24 * 900 + 100 - cs, st
    assert o.size() == cs
    for i in range(100):                                   # touched by every batch: never the least recently used
        a, b_ = o.get_item(f"steady_{i}", now), e.get_item(f"steady_{i}", now)
        assert a is not None and b_ is not None and a["remaining"] == b_["remaining"] == 50 - 24, (i, a, b_)
    e.close()


@pytest.mark.parametrize("flags", [0, 2, 64])
def test_evicted_keys_that_return_meet_the_reference_list(flags):
    """The bounded cache IS the reference's list (lrucache.go:88-149): the victim is the item at the back of the exact recency
    order, it goes in the middle of the batch — at the request whose insert overflows the cache — and a key evicted by request i is
    a new item for request j > i (guber_kernels_lru.h: the eviction pre-pass).  Adversarial workloads over 2 600 keys and a cache of
    2 000, batches of 1 550 in which evicted keys come back in the same and in the next batch: a cyclic scan (the classic worst
    case: every access of an exact LRU misses), a random walk over a working set just above the cache, Zipf, and short durations
    with the clock moving (an expired item keeps its place in the list until somebody asks for it or it reaches the back).  Every
    answer, the size after every batch and gubernator_unexpired_evictions_count equal the bounded-LRU oracle."""
    cs, nkeys, bsz = 2000, 2600, 1500
    rng = np.random.default_rng(3)
    z = streams.ZipfSampler(nkeys, seed=9)
    for name in ("cyclic scan", "random walk", "zipf", "expiring"):
        o, e = Oracle(cache_size=cs), engine(cache_size=cs, max_batch=2048, flags=flags)
        now, pos = streams.NOW0, 0
        for step in range(24):
            if name == "cyclic scan":
                ids = (pos + np.arange(bsz)) % nkeys
                pos += bsz
            elif name == "zipf":
                ids = z.draw(bsz)
            else:
                ids = rng.integers(0, nkeys, bsz)
            keys = [f"ret_{int(i)}" for i in ids] + [f"pin_{i}" for i in range(50)]      # pinned keys: touched by every batch
            b = HostBatch(keys, 1, 1000, 1500 if name == "expiring" else 3_600_000, now,
                          algorithm=np.concatenate([(ids & 1).astype(np.uint8), np.zeros(50, np.uint8)]))
            got, want = e.eval(b), o.eval(b)
            support.assert_results_equal(got, want, f"{name} step {step}")
            assert got.counters() == want.counters() and want.counters()[4] <= cs, (name, step, got.counters(), want.counters())
            now += 1000
        st = e.stats()
        assert st["unexpired_evictions"] == o.counters()[3] and st["cache_size"] == o.size(), (name, st)
        e.close()


@pytest.mark.parametrize("flags", [0, 64])
@pytest.mark.parametrize("precomputed", [True, False], ids=["host_calendar_values", "calendar_on_the_device"])
def test_a_new_key_whose_requests_all_fail_in_the_algorithm_is_no_insert(flags, precomputed):
    """A request with DURATION_IS_GREGORIAN and a duration that is no interval constant (or GregorianWeeks: not supported) fails in
    tokenBucketNewItem / leakyBucketNewItem BEFORE c.Add (algorithms.go; interval.go:93,97,107,125,130,148): for a key that is not in
    the cache it is a GetItem miss and nothing else — no insert, nobody leaves the list for it (lrucache.go:98-100); a key with a
    failing request first and a good one later in the batch is inserted by the good one.  Under a binding cache every answer, the
    counters and the size after every batch equal the bounded-LRU oracle (round 5: the eviction pre-pass counted such keys as inserts
    and evicted one item too many for each — this test fails without lru_cannot_insert, guber_kernels_lru.h)."""
    cs, nkeys, bsz = 2000, 2600, 1500
    rng = np.random.default_rng(17)
    o, e = Oracle(cache_size=cs), engine(cache_size=cs, max_batch=2048, flags=flags)
    now = streams.NOW0
    for step in range(16):
        ids = rng.integers(0, nkeys, bsz)
        keys = [f"lru_{int(i)}" for i in ids]
        beh = np.zeros(bsz, np.uint32); dur = np.full(bsz, 3_600_000, np.int64)
        bad = rng.choice(bsz, 120, replace=False)
        for q, i in enumerate(bad):                              # keys nobody else asks for: all their requests fail
            keys[i] = f"never_{step}_{q % 40}"
            beh[i] = support.GREGORIAN; dur[i] = 3 if q % 2 else 99
        late = rng.choice(np.setdiff1d(np.arange(bsz), bad), 60, replace=False)
        for q, i in enumerate(sorted(late)):                     # a failing request first, a good one for the same key later
            keys[i] = f"late_{step}_{q // 2}"
            if q % 2 == 0:
                beh[i] = support.GREGORIAN; dur[i] = 77
        ge, gd = np.zeros(bsz, np.int64), np.zeros(bsz, np.int64)
        for i in np.nonzero(beh & support.GREGORIAN)[0]:
            ge[i], gd[i] = support.gregorian(now, int(dur[i]))
        algo = (ids & 1).astype(np.uint8)
        want = o.eval(HostBatch(keys, 1, 1000, dur, now, algorithm=algo, behavior=beh, greg_expire=ge, greg_duration=gd))
        got = e.eval(HostBatch(keys, 1, 1000, dur, now, algorithm=algo, behavior=beh, greg_expire=ge, greg_duration=gd) if precomputed
                     else HostBatch(keys, 1, 1000, dur, now, algorithm=algo, behavior=beh))
        support.assert_results_equal(got, want, f"step {step}")
        assert (got.err[bad] != 0).all()
        assert got.counters() == want.counters() and want.counters()[4] <= cs, (step, got.counters(), want.counters())
        now += 1000
    st = e.stats()
    assert st["unexpired_evictions"] == o.counters()[3] and st["cache_size"] == o.size() and st["eviction_passes"] >= 1, st
    e.close()


@pytest.mark.parametrize("flags", [0, 64])
@pytest.mark.parametrize("kinds", ["greg", "reset", "reset+greg"])
def test_requests_that_change_the_lists_length_are_evaluated_on_their_own(flags, kinds):
    """VERDICT r05 item 3 (row A8).  Under a binding cache the eviction pre-pass models every access as "the key is at the front now"
    (guber_kernels_lru.h).  Two kinds of request are not that: a TOKEN_BUCKET RESET_REMAINING of a key in the cache removes the item and
    inserts nothing (algorithms.go:78-90), and a resident key's FIRST request that fails before c.Add (DURATION_IS_GREGORIAN with no
    interval constant) is no arrival at the front if the key had been pushed out before it (round 5's documented divergence: the key
    among the oldest, evicted by the batch's earlier inserts, GetItem misses, nothing is inserted) or had expired (lrucache.go:111-128
    removes it).  The pre-pass reports the first such request (LRU_SPLIT) and the engine evaluates the requests before it, it alone,
    and the rest as batches of their own.  Every answer, the counters and the size after EVERY batch — i.e. which item the next batch
    finds evicted — equal the bounded-LRU oracle's; all six cases fail without the split (checked on the kernel source on the CPU,
    tests/test_kernels_devsim.py)."""
    cs, nkeys, bsz = 2000, 2600, 1500
    o, e = Oracle(cache_size=cs), engine(cache_size=cs, max_batch=2048, flags=flags)
    for step, b in enumerate(streams.length_changing_batches(23, 12, nkeys, bsz, kinds, support.gregorian)):
        want, got = o.eval(b), e.eval(b)
        support.assert_results_equal(got, want, f"{kinds} step {step}")
        assert got.counters() == want.counters() and want.counters()[4] <= cs, (step, got.counters(), want.counters())
    st = e.stats()
    assert st["unexpired_evictions"] == o.counters()[3] and st["cache_size"] == o.size() and st["eviction_passes"] >= 1 and st["batch_cuts"] >= 1, st
    e.close()


def test_a_batch_larger_than_the_cache_is_evaluated_in_pieces():
    """cache_size 300 under batches of 1 000 requests over 500 keys (and one of 70 000 through the radix pipeline's size class): the
    engine cuts the batch into pieces of cache_size requests, each with its own eviction pre-pass — a key evicted by request i is a
    new item for request j > i of the same batch (lrucache.go:98-100), element-wise equal to the oracle; so are a handful of
    requests on the one-launch path's size class and the items AddCacheItem / GetCacheItem see afterwards."""
    cs = 300
    o, e = Oracle(cache_size=cs), engine(cache_size=cs, max_batch=1 << 17)
    rng = np.random.default_rng(5)
    now = streams.NOW0
    for step, n in enumerate([1000, 1000, 40, 1000, 7, 70_000, 1000]):
        ids = rng.integers(0, 500, n)
        b = HostBatch([f"cut_{int(i)}" for i in ids], 1, 50, 3_600_000, now)
        got, want = e.eval(b), o.eval(b)
        support.assert_results_equal(got, want, f"step {step}")
        assert got.counters() == want.counters() and want.counters()[4] == cs, (step, got.counters(), want.counters())
        now += 10
    # LRUCache.Add beyond the size: the oldest items go, whatever is left is the reference's (lrucache.go:88-103)
    items = [support.make_item(f"added_{i}", 0, limit=10, duration=60_000, remaining=10 - (i % 7), stamp=now, expire_at=now + 60_000) for i in range(120)]
    for it in items:
        o.add_item(it, now)
    e.add_item(items[0], now)                                          # (sets the clock Add's eviction classifies expired items against)
    e.add_items(items[1:])
    assert e.size() == o.size() == cs
    probe = HostBatch([f"cut_{i}" for i in range(500)] + [f"added_{i}" for i in range(120)], 0, 50, 3_600_000, now + 1)
    support.assert_results_equal(e.eval(probe), o.eval(probe), "after Add")
    assert e.stats()["unexpired_evictions"] == o.counters()[3]
    e.close()


@pytest.mark.parametrize("flags", [0, 2, 64])
def test_an_invalid_algorithm_request_does_not_refresh_recency(flags):
    """workers.go:317-321 rejects a request with an unknown algorithm BEFORE tokenBucket / leakyBucket call cache.GetItem: it neither
    counts as a cache access nor moves its key to the front of the list (lrucache.go:111-128).  Over a binding cache the difference
    shows as WHO is evicted next: keys that were only "touched" by invalid requests are the oldest and go first.  (Round 4 stamped the
    bucket with the run's last request whatever it was.)  Uniform runs of invalid requests, a segment that mixes valid and invalid ones
    (walked: stamped with its last VALID request), keys that are not resident at all — element-wise equal to the bounded-LRU oracle,
    sizes and counters too, on every batch pipeline."""
    cs = 600
    o, e = Oracle(cache_size=cs), engine(cache_size=cs, max_batch=2048, flags=flags)
    rng = np.random.default_rng(21)
    now = streams.NOW0
    for step in range(20):
        ids = rng.integers(0, 900, 700)
        algo = (ids & 1).astype(np.uint8)
        bad = rng.random(700) < 0.3                                   # a third of the requests carry an algorithm the reference rejects
        if step % 4 == 1:
            bad |= ids % 3 == 0                                       # ... whole keys' runs among them
        algo[bad] = 7
        b = HostBatch([f"inv_{int(i)}" for i in ids], 1, 40, 3_600_000, now, algorithm=algo)
        got, want = e.eval(b), o.eval(b)
        support.assert_results_equal(got, want, f"step {step}")
        assert got.counters() == want.counters(), (step, got.counters(), want.counters())
        now += 500
    assert e.stats()["unexpired_evictions"] == o.counters()[3] and e.size() == o.size()
    e.close()


def test_duplicates_inside_one_add_keep_the_calls_order():
    """LRUCache.Add item by item (lrucache.go:88-103, workers.go:566-581): with [A, B, A', C, B', A'', D, E] in one call the last
    value of a key wins and the recency order is that of the keys' LAST places in the call.  guber_add_items applies duplicates in
    several launches; every item carries the recency number of its place in the CALL, so the next evictions take the reference's
    victims (round 4 numbered the items launch by launch: the re-added keys ended up in front of D and E)."""
    cs = 8
    now = streams.NOW0
    mk = lambda k, rem: support.make_item(k, 0, limit=10, duration=3_600_000, remaining=rem, stamp=now, expire_at=now + 3_600_000)
    o, e = Oracle(cache_size=cs), engine(cache_size=cs, max_batch=1024)
    call = [mk("d_A", 9), mk("d_B", 8), mk("d_A", 7), mk("d_C", 6), mk("d_B", 5), mk("d_A", 4), mk("d_D", 3), mk("d_E", 2)]
    e.add_item(mk("d_seed", 1), now); o.add_item(mk("d_seed", 1), now)        # (also sets the engine's clock)
    for it in call:
        o.add_item(it, now)
    e.add_items(call)
    assert e.size() == o.size() == 6
    # oldest -> newest: seed, C (place 3), B (4), A (5), D (6), E (7).  Five new keys over a cache of 8 holding 6: seed, C and B go
    b = HostBatch([f"d_new{i}" for i in range(5)], 1, 10, 3_600_000, now + 1)
    support.assert_results_equal(e.eval(b), o.eval(b), "new keys")
    assert e.size() == o.size() == cs
    left = {}
    for k in ("d_seed", "d_C", "d_B", "d_A", "d_D", "d_E"):
        a, g = o.get_item(k, now + 2), e.get_item(k, now + 2)
        assert (a is None) == (g is None), (k, a, g)
        left[k] = None if a is None else a["remaining"]
        assert a is None or a["remaining"] == g["remaining"], (k, a, g)
    assert left == {"d_seed": None, "d_C": None, "d_B": None, "d_A": 4, "d_D": 3, "d_E": 2}, left
    e.close()


def test_global_engine_keeps_serving_across_rebuilds():
    """An engine created with GUBER_FLAG_GLOBAL whose directory fills with the entries of expired keys: the table is rebuilt
    (pending GLOBAL records move with their buckets) instead of rejecting the batch, and what was queued before the rebuild
    is still delivered by guber_global_take."""
    e = engine(cache_size=512, max_batch=1024, max_key_bytes=64, table_slots=4096, flags=ga.FLAG_GLOBAL)     # 4096 slots, 3584 tags
    o = Oracle(cache_size=1 << 20)
    now = streams.NOW0
    gk = [f"glob_{i}" for i in range(40)]
    b = HostBatch(gk, 3, 100, 3_600_000, now, behavior=support.GLOBAL, is_owner=0)
    support.assert_results_equal(e.eval(b), o.eval(b), "global batch")
    for step in range(30):                                 # 30 x 400 short-lived keys >> 3584 directory entries
        keys = [f"shortlived_{step}_{i}" for i in range(400)]
        b = HostBatch(keys, 1, 5, 20, now)
        got = e.eval(b)
        support.assert_results_equal(got, o.eval(b), f"step {step}")
        now += 1000
    st = e.stats()
    assert st["compactions"] >= 2 and st["tags_used"] < 3584, st
    rows = e.global_take(2)
    assert len(rows) == 40 and sorted(rows.keys()) == sorted(k.encode() for k in gk) and (rows.hits == 3).all()
    e.close()


def test_long_key_arena_is_reclaimed_by_rebuilds():
    """Keys above 62 bytes live in an append-only arena; the rebuild copies the live ones into a fresh arena, so a churning
    population of long keys never exhausts it."""
    o, e = Oracle(cache_size=1 << 20), engine(cache_size=2048, max_batch=1024, max_key_bytes=200, table_slots=4096)    # 1 MiB arena, 3584 tags
    now = streams.NOW0
    for step in range(40):                                  # 40 x 600 x 160 B = 3.8 MB of key bytes through a 1 MiB arena
        keys = [f"long_{step}_{i}_" + "x" * 140 for i in range(600)] + ["resident_" + "y" * 100 + str(i) for i in range(50)]
        b = HostBatch(keys, 1, 9, [30] * 600 + [3_600_000] * 50, now)
        got = e.eval(b)
        support.assert_results_equal(got, o.eval(b), f"step {step}")
        assert (got.err[:b.n] == 0).all(), step
        now += 1000
    assert e.stats()["compactions"] >= 3
    e.close()


def test_extreme_value_runs_on_the_device():
    """Go's wrap-around and float->int rules on the device: int64 extremes / negatives for every request field, items
    pre-loaded with extreme Remaining (token int64, leaky float64 incl. fractions, negatives, 1e300), runs of identical
    requests on one key — rank by rank and state by state equal to the oracle (CPU twin: test_kernel_logic_host.py)."""
    rng = np.random.default_rng(2025)
    now = streams.NOW0
    I64 = [0, 1, -1, 2, 3, 7, 100, 2**31, 2**53, 2**53 + 1, 2**62, -(2**62), 2**63 - 1, -(2**63), 2**63 - 2, -(2**63) + 1]
    F64 = [0.0, 0.5, 1.0, 1.5, -1.0, -0.25, 99.999, 2.0**53, 2.0**53 + 2, 9.3e18, -9.3e18, 1e300, -1e300, 3.0, 10.0]
    o, e = Oracle(cache_size=1 << 16), engine(cache_size=1 << 14, max_batch=1024)
    for trial in range(400):
        key = b"xk_%d" % trial
        t = now + int(rng.integers(0, 10_000))
        algo0 = int(rng.integers(0, 2))
        if rng.random() < 0.7:
            it = dict(limit=int(rng.choice(I64)), duration=int(rng.choice([1000, 60_000, 0, -5, 2**62])), remaining=int(rng.choice(I64)),
                      remaining_f=float(rng.choice(F64)), stamp=t - int(rng.choice([0, 1, 999, 10**9])), burst=int(rng.choice(I64[:8] + [2**62])),
                      expire_at=t + int(rng.choice([0, 1, 60_000, -1, 2**62])))
            for be in (o, e):
                be.add_item(support.make_item(key, algo0, **it), t)
        for phase in range(int(rng.integers(1, 4))):
            n = int(rng.choice([1, 2, 3, 5, 40, 200]))
            hits = int(rng.choice(I64 + [1, 1, 1, 2, 5]))
            limit = int(rng.choice(I64 + [10, 100]))
            duration = int(rng.choice([0, 1, 3, 1000, 60_000, -1, -(2**62), 2**62, 2**63 - 1]))
            algo = int(rng.choice([0, 1]))
            beh = int(rng.choice([0, 0, 32, 8, 40]))
            burst = int(rng.choice([0, 0, 15, -3, 2**62, 2**63 - 1]))
            created = int(rng.choice([t, t, t - 5, t + 5, 0, -1, 2**62, -(2**62), t - 10**9]))
            b = HostBatch([key] * n, hits, limit, duration, t, burst=burst, created_at=created, algorithm=algo, behavior=beh)
            support.assert_results_equal(e.eval(b), o.eval(b), f"trial {trial} phase {phase} algo {algo} hits {hits} limit {limit} dur {duration} "
                                                            f"burst {burst} created {created} beh {beh}")
            a, b_ = o.get_item(key, t), e.get_item(key, t)
            if a is None or b_ is None:
                assert a is None and b_ is None, (trial, phase)
            else:
                for f in ("algorithm", "status", "limit", "duration", "remaining", "stamp", "burst", "expire_at"):
                    assert a[f] == b_[f], (trial, phase, f, a, b_)
                assert a["remaining_f"] == b_["remaining_f"] or (a["remaining_f"] != a["remaining_f"] and b_["remaining_f"] != b_["remaining_f"]), (trial, a, b_)
            t += int(rng.choice([0, 1, 40, 1200, 70_000]))
    e.close()


def test_stages_of_several_engines_in_one_submission():
    """guber_stages_submit (what the pool's dispatcher calls): stages of three engines on one stream — an empty one, batches of 5
    and 200 requests (one k_small_multi), 300, 3000 and 40 000 requests (copy kernel + fused two-launch pipeline) — never blocks,
    guber_stage_poll reports completion, every answer equals the oracle's, and a second submission of the same engines keeps
    per-key order."""
    import time
    rng = np.random.default_rng(12)
    now = streams.NOW0
    first = engine(cache_size=1 << 17, max_batch=65536)
    engines = [first] + [engine(cache_size=1 << 17, max_batch=65536, stream=first.stream_handle()) for _ in range(2)]
    oracles = [Oracle(cache_size=1 << 18) for _ in engines]
    stages = [[ga.Stage(e, 65536, key_bytes_cap=65536 * 24) for _ in range(2)] for e in engines]
    sizes = [(0, 5, 300), (200, 3000, 40_000), (40_000, 7, 0), (1, 1, 1)]
    for rnd, ns in enumerate(sizes):
        batch = []
        for j, n in enumerate(ns):
            ids = rng.zipf(1.3, n) % 5000 if n else np.zeros(0, np.int64)
            hb = HostBatch([f"st{j}_k{int(i)}" for i in ids], rng.integers(0, 3, n), 40, 5_000, now + rnd * 2_600, algorithm=(ids % 2).astype(np.uint8),
                           created_at=now + rnd * 2_600)
            stages[j][rnd % 2].fill(hb)
            batch.append(hb)
        assert ga.Stage.submit_many([stages[j][rnd % 2] for j in range(3)]) == 3
        t0 = time.time()
        while not all(stages[j][rnd % 2].poll() for j in range(3)):
            assert time.time() - t0 < 10
        for j, hb in enumerate(batch):
            stages[j][rnd % 2].wait()
            if hb.n:
                support.assert_results_equal(stages[j][rnd % 2].result(), oracles[j].eval(hb), f"round {rnd} engine {j}")
    assert sum(e.stats()["fused_batches"] for e in engines) >= 2 and sum(e.stats()["small_batches"] for e in engines) >= 4
    for j, e in enumerate(engines):
        assert e.size() == oracles[j].size()
    for row in stages:
        for st in row:
            st.close()
    for e in reversed(engines):                                               # (the engine that owns the stream goes last)
        e.close()


def test_a_generation_of_nine_engines_in_one_pair_of_launches():
    """guber_stages_submit with more stages than fit the kernel-argument segment (a pool dispatcher's generation over 8 shards
    and the GLOBAL engine): the argument blocks travel through device memory, ONE completion event serves the group — 40
    generations (more than the ring of group events holds) alternating two stages per engine, sizes ragged between 257 and
    5000 requests with a small batch and an empty stage mixed in, time moving so buckets renew: every answer equals the
    oracle's and all but the small / empty stages went through fused launches."""
    import time
    rng = np.random.default_rng(77)
    now = streams.NOW0
    first = engine(cache_size=1 << 16, max_batch=8192)
    engines = [first] + [engine(cache_size=1 << 16, max_batch=8192, stream=first.stream_handle()) for _ in range(8)]
    oracles = [Oracle(cache_size=1 << 17) for _ in engines]
    stages = [[ga.Stage(e, 8192, key_bytes_cap=8192 * 24) for _ in range(2)] for e in engines]
    fused_expected = 0
    for rnd in range(40):
        batch = []
        t = now + rnd * 700
        for j in range(9):
            n = int(rng.integers(257, 5000))
            if (rnd + j) % 11 == 0: n = 0
            if (rnd + j) % 7 == 3: n = int(rng.integers(1, 200))
            ids = rng.zipf(1.2, n) % 3000 if n else np.zeros(0, np.int64)
            hb = HostBatch([f"g{j}_k{int(i)}" for i in ids], rng.integers(0, 3, n), 25, 2_000, t, algorithm=(ids % 2).astype(np.uint8), created_at=t)
            stages[j][rnd % 2].fill(hb)
            batch.append(hb)
            fused_expected += n > 256
        assert ga.Stage.submit_many([stages[j][rnd % 2] for j in range(9)]) == 9
        t0 = time.time()
        while not all(stages[j][rnd % 2].poll() for j in range(9)):
            assert time.time() - t0 < 10
        for j, hb in enumerate(batch):
            stages[j][rnd % 2].wait()
            if hb.n:
                support.assert_results_equal(stages[j][rnd % 2].result(), oracles[j].eval(hb), f"generation {rnd} engine {j}")
    assert sum(e.stats()["fused_batches"] for e in engines) >= fused_expected - 40       # (a generation with one large stage is not "fused")
    for j, e in enumerate(engines):
        assert e.size() == oracles[j].size()
    for row in stages:
        for st in row:
            st.close()
    for e in reversed(engines):
        e.close()


def test_one_stage_routed_to_nine_engines():
    """guber_stage_submit_routed (the device-level stage of a pool): requests in arrival order, each tagged with its engine and
    its rank there; the device places the shares, runs them as ONE pair of launches and answers in arrival order.  30 batches
    over 9 engines (ragged: 1 .. 20 000 requests, variable-length keys, token and leaky, hits 0..2, an engine with no share,
    an engine with one request, duplicate keys inside and across batches, time moving so buckets renew and expire) — every
    answer equals what that engine's oracle says for its share, table sizes agree, and a plain batch on one of the engines
    afterwards still agrees (the engines' work arrays were left consistent)."""
    import time
    rng = np.random.default_rng(2024)
    now = streams.NOW0
    first = engine(cache_size=1 << 16, max_batch=32768)
    engines = [first] + [engine(cache_size=1 << 16, max_batch=32768, stream=first.stream_handle()) for _ in range(8)]
    oracles = [Oracle(cache_size=1 << 17) for _ in engines]
    stages = [ga.Stage(first, 32768, key_bytes_cap=32768 * 40) for _ in range(2)]
    sizes = [1, 2, 9, 255, 256, 257, 1000, 5000, 20_000, 3, 700, 12_000]
    for rnd in range(30):
        n = sizes[rnd % len(sizes)]
        t = now + rnd * 450
        ids = rng.zipf(1.15, n) % 6000
        keys = [f"acct_{int(i)}" + "x" * int(i % 23) for i in ids]                  # variable-length keys
        shard = (ids * 2654435761 % 9).astype(np.uint32)                            # a key always goes to the same engine
        if rnd % 5 == 1: shard[shard == 4] = 5                                      # ... except when a "placement pass" empties engine 4 for a batch
        hb = HostBatch(keys, rng.integers(0, 3, n), 30, 3_000, t, algorithm=(ids % 2).astype(np.uint8), created_at=t, burst=np.where(ids % 2 == 1, 40, 0))
        st = stages[rnd % 2]
        st.fill(hb)
        counts = st.submit_routed(engines, shard)
        t0 = time.time()
        while not st.poll():
            assert time.time() - t0 < 10
        st.wait()
        got = st.result()
        for j in range(9):
            idx = np.nonzero(shard == j)[0]
            assert len(idx) == counts[j]
            if not len(idx):
                continue
            sub = HostBatch([keys[i] for i in idx], hb.hits[idx], 30, 3_000, t, algorithm=hb.algorithm[idx], created_at=t, burst=hb.burst[idx])
            want = oracles[j].eval(sub)
            for f in ("status", "err", "limit", "remaining", "reset_time"):
                a, b = getattr(got, f)[idx], getattr(want, f)[:len(idx)]
                assert np.array_equal(a, b), f"round {rnd} engine {j} field {f}: first difference at {idx[np.nonzero(a != b)[0][:5]]}"
    for j, e in enumerate(engines):
        assert e.size() == oracles[j].size(), j
    hb = HostBatch([f"acct_{i}" + "x" * (i % 23) for i in range(3000)], 1, 30, 3_000, now + 30 * 450, algorithm=(np.arange(3000) % 2).astype(np.uint8), created_at=now + 30 * 450)
    sel = [i for i in range(3000) if i * 2654435761 % 9 == 2]
    sub = HostBatch([hb_key for hb_key in (f"acct_{i}" + "x" * (i % 23) for i in sel)], 1, 30, 3_000, now + 30 * 450, algorithm=(np.array(sel) % 2).astype(np.uint8),
                    created_at=now + 30 * 450)
    support.assert_results_equal(engines[2].eval(sub), oracles[2].eval(sub), "a plain batch after the routed ones")
    for st in stages:
        st.close()
    for e in reversed(engines):
        e.close()


def test_a_routed_stage_with_faulty_ranks_is_refused_not_run():
    """guber_stage_dest is written by the caller.  A share of a <= 256-request routed stage whose ranks are not a permutation of
    0..count-1 (a rank twice, a rank past the share) is not evaluated — the wait reports GUBER_E_INVALID_ARG, nothing outside the
    stage is touched, the engines keep working and a later well-formed stage on the same engines agrees with the oracle."""
    first = engine(cache_size=1 << 12, max_batch=4096)
    engines = [first] + [engine(cache_size=1 << 12, max_batch=4096, stream=first.stream_handle()) for _ in range(2)]
    oracles = [Oracle(cache_size=1 << 13) for _ in engines]
    st = ga.Stage(first, 4096, key_bytes_cap=4096 * 24)
    now = streams.NOW0
    keys = [f"rt_{i % 37}" for i in range(120)]
    shard = (np.arange(120) % 37 % 3).astype(np.uint32)

    def twice(dest):                                    # two requests of engine 1 claim the same rank
        mine = np.nonzero((dest >> 24) == 1)[0]
        dest[mine[3]] = dest[mine[2]]

    def past(dest):                                     # a rank past the end of engine 2's share
        mine = np.nonzero((dest >> 24) == 2)[0]
        dest[mine[0]] = (2 << 24) | 200

    for fault in (twice, past):
        hb = HostBatch(keys, 1, 50, 60_000, now, created_at=now)
        st.fill(hb)
        st.submit_routed(engines, shard, corrupt_dest=fault)
        with pytest.raises(ga.GuberError) as ei:
            st.wait()
        assert ei.value.code == ga.E_INVALID_ARG, ei.value
    # the shares of the well-formed engines of those two stages did run (2 hits each on engine 0); a clean stage sees exactly that
    for rnd in range(3):
        hb = HostBatch(keys, 1, 50, 60_000, now + 1 + rnd, created_at=now + 1 + rnd)
        st.fill(hb)
        st.submit_routed(engines, shard)
        st.wait()
        got = st.result()
        idx = np.nonzero(shard == 0)[0]
        sub = HostBatch([keys[i] for i in idx], 1, 50, 60_000, now + 1 + rnd, created_at=now + 1 + rnd)
        if rnd == 0:
            for t in (now, now):
                oracles[0].eval(HostBatch([keys[i] for i in idx], 1, 50, 60_000, t, created_at=t))
        want = oracles[0].eval(sub)
        for f in ("status", "err", "limit", "remaining", "reset_time"):
            assert np.array_equal(getattr(got, f)[idx], getattr(want, f)[:len(idx)]), (rnd, f)
    st.close()
    for e in reversed(engines):
        e.close()


def _host_routing(pl, key_bytes, key_off, behavior, n_engines, global_engine):
    """what the pool's callers computed before the device did: shard by the placement (or the GLOBAL engine), rank = arrival order"""
    shard, _ = pl.route_keys(key_bytes, key_off)
    shard = shard.astype(np.uint32).copy()
    if global_engine >= 0:
        shard[(behavior & 2) != 0] = global_engine
    counts = np.bincount(shard, minlength=n_engines).astype(np.uint32)
    order = np.argsort(shard, kind="stable")
    start = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
    rank = np.empty(len(shard), np.uint32)
    rank[order] = (np.arange(len(shard), dtype=np.int64) - start[shard[order]]).astype(np.uint32)
    return (shard << 24) | rank, counts


def test_the_device_routes_a_front_stage_like_the_placement():
    """guber_stage_route (k_route_count + k_route_dest): WorkerPool.getWorker (workers.go:153-155, 180-184) generalised by the placement's
    slot table and hot-key list, for a whole stage on the device.  dest and the shares' sizes equal what the host computes with
    guber_placement_route_keys + arrival-order ranks: before any traffic was observed (the reference's contiguous ranges), after
    a rebalance that placed slots afresh and pinned hot keys, with GLOBAL requests going to the GLOBAL engine; sizes 1 .. 65 536,
    variable-length keys; and the stage then runs through guber_stage_submit_routed and agrees with the oracles."""
    rng = np.random.default_rng(77)
    n_plain, glob = 7, 7
    first = engine(cache_size=1 << 16, max_batch=65536)
    engines = [first] + [engine(cache_size=1 << 16, max_batch=65536, stream=first.stream_handle()) for _ in range(n_plain)]
    oracles = [Oracle(cache_size=1 << 17) for _ in engines]
    st = ga.Stage(first, 65536, key_bytes_cap=65536 * 40)
    pl = ga.Placement(n_plain)
    now = streams.NOW0
    rule = pl.export(global_engine=glob)
    for rnd, n in enumerate([1, 255, 256, 257, 4000, 65536, 20_000, 3, 30_000]):
        if rnd == 4:                                                 # traffic observed -> slots placed afresh, hot keys pinned
            ids = rng.zipf(1.1, 200_000) % 50_000
            seen = HostBatch([f"acct_{int(i)}" + "y" * int(i % 19) for i in ids], 1, 1, 1, now)
            pl.observe_keys(seen.key_bytes, seen.key_off)
            pl.rebalance(0.05, move_slots=True)
            assert pl.n_hot() > 0
            rule = pl.export(global_engine=glob)
        t = now + rnd * 500
        ids = rng.zipf(1.1, n) % 50_000
        keys = [f"acct_{int(i)}" + "y" * int(i % 19) for i in ids]
        beh = np.where(rng.random(n) < 0.1, 2, 0).astype(np.uint32) | np.where(rng.random(n) < 0.2, 1, 0).astype(np.uint32)
        hb = HostBatch(keys, rng.integers(0, 3, n), 40, 5_000, t, algorithm=(ids % 2).astype(np.uint8), created_at=t, behavior=beh, burst=np.where(ids % 2 == 1, 50, 0))
        st.fill(hb)
        dest, counts = st.route(rule if rnd in (0, 4) else None, len(engines))
        want_dest, want_counts = _host_routing(pl, hb.key_bytes, hb.key_off, beh, len(engines), glob)
        assert np.array_equal(counts, want_counts), (rnd, counts, want_counts)
        bad = np.nonzero(dest != want_dest)[0]
        assert len(bad) == 0, (rnd, n, bad[:5], dest[bad[:5]], want_dest[bad[:5]])
        st.submit_routed_as_routed(engines, counts)
        st.wait()
        got = st.result()
        shard = dest >> 24
        for j in range(len(engines)):
            idx = np.nonzero(shard == j)[0]
            if not len(idx):
                continue
            sub = HostBatch([keys[i] for i in idx], hb.hits[idx], 40, 5_000, t, algorithm=hb.algorithm[idx], created_at=t, behavior=beh[idx], burst=hb.burst[idx])
            want = oracles[j].eval(sub)
            for f in ("status", "err", "limit", "remaining", "reset_time"):
                assert np.array_equal(getattr(got, f)[idx], getattr(want, f)[:len(idx)]), (rnd, j, f)
    st.close()
    pl.close()
    for e in reversed(engines):
        e.close()


def test_internal_retries_follow_the_order_of_the_waits():
    """include/guber_gpu.h, stages: items that hit the internal retry (GUBER_ITEM_E_RETRY: a key whose 64-bit hash is another key's) are
    re-run by the guber_stage_wait of THEIR stage, i.e. after whatever is already in flight.  A later stage's requests of such a key
    meet the same resident key, are retried as well and re-run by THEIR stage's wait — so per key the order is the order of the
    waits.  Pinned with the weak test hash (900 new keys under 64 hash values: most of the first batch retries), two stages in flight:
      * waits in submission order: element-wise equal to the oracle evaluating stage one, then stage two;
      * waits in the opposite order: every request still answered exactly once (per key the multiset of `remaining` is the
        oracle's, the bucket ends where the oracle's ends, a stage's own requests of a key keep their order) and second-stage
        requests were applied BEFORE retried first-stage ones — the reorder the header documents, which two concurrent GetRateLimits
        calls have in the reference as well (gubernator.go:183-306 orders nothing between RPCs)."""
    n = 3000
    idx_a, idx_b = np.arange(n) % 900, (np.arange(n) * 7) % 900
    now = streams.NOW0
    ha = HostBatch([f"retry_{k}" for k in idx_a], 1, 1000, 600_000, now, created_at=now)
    hb = HostBatch([f"retry_{k}" for k in idx_b], 1, 1000, 600_000, now, created_at=now)
    for in_order in (True, False):
        eng = engine(cache_size=1 << 14, max_batch=4096, flags=ga.FLAG_TEST_WEAK_HASH)
        sa, sb = ga.Stage(eng, 4096), ga.Stage(eng, 4096)
        sa.fill(ha); sb.fill(hb)
        sa.submit(); sb.submit()
        for st in ((sa, sb) if in_order else (sb, sa)):
            st.wait()
        ra, rb = sa.result(), sb.result()
        assert eng.stats()["retries"] > 0, "the weak hash produced no internal retry: the test shows nothing"
        orc = Oracle(cache_size=1 << 15)
        wa, wb = orc.eval(ha), orc.eval(hb)
        if in_order:
            support.assert_results_equal(ra, wa, "first stage (two in flight, waits in submission order)")
            support.assert_results_equal(rb, wb, "second stage (two in flight, waits in submission order)")
        else:
            assert not ra.err.any() and not rb.err.any()
            overtaken = 0
            for k in range(900):
                ia, ib = np.nonzero(idx_a == k)[0], np.nonzero(idx_b == k)[0]
                got = np.sort(np.concatenate([ra.remaining[ia], rb.remaining[ib]]))
                want = np.sort(np.concatenate([wa.remaining[ia], wb.remaining[ib]]))
                assert np.array_equal(got, want), (k, got, want)                 # each request applied exactly once
                assert np.all(np.diff(ra.remaining[ia]) < 0) and np.all(np.diff(rb.remaining[ib]) < 0), k
                if len(ia) and len(ib) and rb.remaining[ib].max() > ra.remaining[ia].min():
                    overtaken += 1
            assert overtaken > 0, "waiting for the second stage first did not re-run its retries first"
            print(f"waits in the opposite order: {overtaken} of 900 keys had second-stage requests applied before retried first-stage ones; retries {eng.stats()['retries']}")
        probe = HostBatch([f"retry_{k}" for k in range(900)], 0, 1000, 600_000, now + 1, created_at=now + 1)
        support.assert_results_equal(eng.eval(probe), orc.eval(probe), "bucket state after both stages")
        sa.close(); sb.close(); eng.close()


def test_buckets_move_between_tables_by_key_hash():
    """guber_move_items_by_hash (a hot key changes its logical shard): token and leaky items, an inline key and a 200-byte key
    (arena), an expired item and a hash that names nothing — the items arrive unchanged, leave nothing behind, and both
    engines keep evaluating exactly like ONE oracle that never heard of the move."""
    import xxhash
    now = streams.NOW0
    a = engine(cache_size=4096, max_batch=1024, max_key_bytes=256)
    b = engine(cache_size=4096, max_batch=1024, max_key_bytes=256, stream=a.stream_handle())
    o = Oracle(cache_size=1 << 16)
    keys = ["mv_token", "mv_leaky", "mv_" + "L" * 197, "mv_expired", "mv_stays"]
    algo = np.array([0, 1, 0, 0, 1], np.uint8)
    hb = HostBatch(keys, 3, 10, [60_000, 60_000, 60_000, 5, 60_000], now, algorithm=algo, created_at=now)
    support.assert_results_equal(a.eval(hb), o.eval(hb), "before")
    hashes = [xxhash.xxh64(k.encode(), seed=0).intdigest() for k in keys[:4]] + [0x1234567890abcdef]
    assert a.move_items_to(b, hashes) == 4
    assert a.size() == 1 and b.size() == 4
    for k in keys[:3]:
        assert a.get_item(k, now + 1) is None
        it = b.get_item(k, now + 1)
        assert it is not None and it["limit"] == 10
    later = now + 10
    moved = HostBatch(keys[:4], 2, 10, [60_000, 60_000, 60_000, 5], later, algorithm=algo[:4], created_at=later)
    stays = HostBatch(keys[4:], 2, 10, 60_000, later, algorithm=algo[4:], created_at=later)
    support.assert_results_equal(b.eval(moved), o.eval(moved), "moved keys on their new table")       # incl. the expired one: renewed, as the oracle does
    support.assert_results_equal(a.eval(stays), o.eval(stays), "the key that stayed")
    assert a.size() + b.size() == o.size()
    b.close(); a.close()
