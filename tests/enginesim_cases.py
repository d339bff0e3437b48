"""Cases of tests/test_enginesim_cpu.py, run in a process of their own with GUBER_HIP_LIB pointing at tests/hostsim/libenginesim.so —
the engine's HOST code and kernels compiled for the CPU (tests/hostsim/enginesim.cpp: test infrastructure, not a fallback; the
product library is hipcc's and needs a device).  "Device" pointers are numpy buffers here.
    GUBER_HIP_LIB=tests/hostsim/libenginesim.so python tests/enginesim_cases.py <case>"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import gubernator_amd as ga
import streams
import support

assert "enginesim" in ga.LIB_PATH, "these cases are for the CPU build of the engine (GUBER_HIP_LIB)"


def dev_batch(hb):
    """a HostBatch's columns as the "device" arrays of a guber_batch_t + result arrays; returns (batch, result, keep-alive, result dict)"""
    n = hb.n
    cols = [np.ascontiguousarray(hb.key_bytes), np.ascontiguousarray(hb.key_off.view(np.int32)), np.ascontiguousarray(hb.hits), np.ascontiguousarray(hb.limit),
            np.ascontiguousarray(hb.duration), np.ascontiguousarray(hb.algorithm), np.ascontiguousarray(hb.behavior.view(np.int32))]
    p = [c.ctypes.data for c in cols]
    r = dict(status=np.empty(n, np.uint8), err=np.empty(n, np.uint8), limit=np.empty(n, np.int64), remaining=np.empty(n, np.int64), reset_time=np.empty(n, np.int64))
    b = ga.GuberBatch(n, 0, p[0], p[1], p[2], p[3], p[4], None, None, p[5], p[6], None, None, None, hb.now_ms)
    res = ga.GuberResult(r["status"].ctypes.data, r["limit"].ctypes.data, r["remaining"].ctypes.data, r["reset_time"].ctypes.data, r["err"].ctypes.data, 0, 0, 0, 0, 0)
    return b, res, cols, r


def routed(n_engines, fuse_ep):
    """guber_eval_batches_routed_dev over n_engines tables on ONE stream, 8 rounds in one call: Zipf batches with hot keys, both
    algorithms, the clock stepping; a batch too small for the owner-partitioned pipeline in one round, a round without one table,
    uniform keys in two rounds (owners split, the owner count moves).  Every batch equals its table's oracle, sizes and counters too.
    With GUBER_FUSE_EP=1: k_evalpart_multi carried most passes; without: it never ran."""
    K, B, rounds = 3000, 2048, 8
    tab = streams.key_table(K * n_engines)
    e0 = ga.Engine(cache_size=1 << 16, max_batch=4 * B)
    engs = [e0] + [ga.Engine(cache_size=1 << 16, max_batch=4 * B, stream=e0.stream_handle()) for _ in range(n_engines - 1)]
    orcs = [support.Oracle(cache_size=1 << 16) for _ in range(n_engines)]
    zs = [streams.ZipfSampler(K, seed=300 + j) for j in range(n_engines)]
    rng = np.random.default_rng(5)
    for e in engs:
        e.profile(True)
    which, hbs, keep, cb, cr = [], [], [], [], []
    for r in range(rounds):
        for j in range(n_engines):
            if r == 4 and j == 2:
                continue
            n = 300 if (r == 2 and j == 1) else [B, B, 1500, B, 2 * B][(r + j) % 5]
            ids = rng.integers(0, K, n) if r in (5, 6) else zs[j].draw(n)
            hb = streams.bench_batch(tab, j * K + ids, streams.NOW0 + r * 900, algorithm=(r + j) % 2, limit=30, duration=4000)
            b, res, cols, rd = dev_batch(hb)
            keep.append((cols, rd)); which.append(j); hbs.append(hb); cb.append(b); cr.append(res)
    N = len(which)
    ga.Engine.eval_routed_dev(engs, (C.c_uint32 * N)(*which), (ga.GuberBatch * N)(*cb), (ga.GuberResult * N)(*cr), N)
    sums = [[0, 0, 0] for _ in range(n_engines)]
    for s in range(N):
        want = orcs[which[s]].eval(hbs[s])
        got = ga.HostResult(hbs[s].n)
        for name in ("status", "limit", "remaining", "reset_time", "err"):
            getattr(got, name)[:] = keep[s][1][name]
        support.assert_results_equal(got, want, f"batch {s} of table {which[s]}")
        for q in range(3):
            sums[which[s]][q] += want.counters()[q]
    launches = {}
    for e in engs:
        for k, v in e.profile_read().items():
            launches[k] = launches.get(k, 0) + v[0]
    print("launches", {k: v for k, v in launches.items() if v})
    groups = (n_engines + 3) // 4
    if fuse_ep:
        assert launches.get("k_evalpart_multi", 0) >= (rounds - 4) * groups, launches
        # every pass has ONE k_own_multi; its k_eval3 went with the next pass's k_part or, held back to the end / flushed, on its own
        assert launches["k_own_multi"] == launches["k_evalpart_multi"] + launches["k_eval3_multi"], launches
        assert launches["k_own_multi"] == launches["k_evalpart_multi"] + launches["k_part_multi"], launches
    else:
        assert launches.get("k_evalpart_multi", 0) == 0 and launches["k_part_multi"] == launches["k_own_multi"] == launches["k_eval3_multi"], launches
    for j, (e, o) in enumerate(zip(engs, orcs)):
        assert e.size() == o.size(), (j, e.size(), o.size())
        assert list(e.counters()[:3]) == sums[j], (j, e.counters(), sums[j])
        e.close()


def routed_with_a_second_thread(fuse_ep):
    """while ONE guber_eval_batches_routed_dev call works through 10 rounds on four tables (holding k_eval3 launches back with
    GUBER_FUSE_EP), another thread keeps calling entry points on two of the tables — guber_size, guber_get_item (read-only, so the
    oracle comparison stays meaningful): whoever takes an engine's mutex launches the evaluation held back for it first
    (guber_engine::held), so the reader never sees a table one batch behind its own stream, nothing deadlocks, and every answer of the
    routed call still equals the oracle"""
    import threading
    n_engines, K, B, rounds = 4, 3000, 2048, 10
    tab = streams.key_table(K * n_engines)
    e0 = ga.Engine(cache_size=1 << 16, max_batch=4 * B)
    engs = [e0] + [ga.Engine(cache_size=1 << 16, max_batch=4 * B, stream=e0.stream_handle()) for _ in range(n_engines - 1)]
    orcs = [support.Oracle(cache_size=1 << 16) for _ in range(n_engines)]
    zs = [streams.ZipfSampler(K, seed=700 + j) for j in range(n_engines)]
    for e in engs:
        e.profile(True)
    which, hbs, keep, cb, cr = [], [], [], [], []
    for r in range(rounds):
        for j in range(n_engines):
            hb = streams.bench_batch(tab, j * K + zs[j].draw(B), streams.NOW0 + r * 900, algorithm=(r + j) % 2, limit=30, duration=4000)
            b, res, cols, rd = dev_batch(hb)
            keep.append((cols, rd)); which.append(j); hbs.append(hb); cb.append(b); cr.append(res)
    N = len(which)
    stop, seen, errors = threading.Event(), [0], []
    probe_key = bytes(tab[1 * K + 5])

    def reader():
        try:
            while not stop.is_set():
                engs[1].size()
                engs[3].get_item(probe_key, streams.NOW0)
                seen[0] += 1
        except Exception as ex:   # noqa: BLE001
            errors.append(repr(ex))

    t = threading.Thread(target=reader)
    t.start()
    try:
        ga.Engine.eval_routed_dev(engs, (C.c_uint32 * N)(*which), (ga.GuberBatch * N)(*cb), (ga.GuberResult * N)(*cr), N)
    finally:
        stop.set()
        t.join()
    assert not errors and seen[0] > 0, (errors, seen)
    for s in range(N):
        want = orcs[which[s]].eval(hbs[s])
        got = ga.HostResult(hbs[s].n)
        for name in ("status", "limit", "remaining", "reset_time", "err"):
            getattr(got, name)[:] = keep[s][1][name]
        support.assert_results_equal(got, want, f"batch {s} of table {which[s]}")
    launches = {}
    for e in engs:
        for k, v in e.profile_read().items():
            launches[k] = launches.get(k, 0) + v[0]
    print("launches", {k: v for k, v in launches.items() if v}, "reader calls", seen[0])
    if fuse_ep:      # (evaluations launched by the reader's calls are not timed: only the dispatcher's own k_eval3_multi are counted)
        assert launches.get("k_evalpart_multi", 0) + launches.get("k_eval3_multi", 0) <= launches["k_own_multi"] == rounds, launches
    for j, (e, o) in enumerate(zip(engs, orcs)):
        assert e.size() == o.size(), (j, e.size(), o.size())
        e.close()


def routed_lru(fuse_ep):
    """four tables whose caches come to BIND (6 000 items each, 7 000 keys in play, uniform draws: after a few rounds): batches that may overflow the cache leave the fused
    groups and go through the eviction pre-pass on their own, the others share launches (and, with GUBER_FUSE_EP, hold their k_eval3
    back) — every answer equals the bounded-LRU oracle's, evictions included"""
    n_engines, K, rounds = 4, 7000, 8
    tab = streams.key_table(K * n_engines)
    e0 = ga.Engine(cache_size=6000, max_batch=4096)
    engs = [e0] + [ga.Engine(cache_size=6000, max_batch=4096, stream=e0.stream_handle()) for _ in range(n_engines - 1)]
    orcs = [support.Oracle(cache_size=6000) for _ in range(n_engines)]
    rng = np.random.default_rng(9)
    zs = [streams.ZipfSampler(K, seed=500 + j) for j in range(n_engines)]
    for e in engs:
        e.profile(True)
    which, hbs, keep, cb, cr = [], [], [], [], []
    for r in range(rounds):
        for j in range(n_engines):
            hb = streams.bench_batch(tab, j * K + (rng.integers(0, K, [1500, 2048, 1100][(r + j) % 3]) if r % 3 else zs[j].draw(1500)), streams.NOW0 + r * 50, algorithm=(r + j) % 2, limit=30, duration=60_000)
            b, res, cols, rd = dev_batch(hb)
            keep.append((cols, rd)); which.append(j); hbs.append(hb); cb.append(b); cr.append(res)
    N = len(which)
    ga.Engine.eval_routed_dev(engs, (C.c_uint32 * N)(*which), (ga.GuberBatch * N)(*cb), (ga.GuberResult * N)(*cr), N)
    for s in range(N):
        want = orcs[which[s]].eval(hbs[s])
        got = ga.HostResult(hbs[s].n)
        for name in ("status", "limit", "remaining", "reset_time", "err"):
            getattr(got, name)[:] = keep[s][1][name]
        support.assert_results_equal(got, want, f"batch {s} of table {which[s]}")
    launches = {}
    for e in engs:
        for k, v in e.profile_read().items():
            launches[k] = launches.get(k, 0) + v[0]
    print("launches", {k: v for k, v in launches.items() if v})
    evicted = 0
    for j, (e, o) in enumerate(zip(engs, orcs)):
        assert e.size() == o.size() <= 6000, (j, e.size(), o.size())
        evicted += e.stats()["unexpired_evictions"] if "unexpired_evictions" in e.stats() else 0
        e.close()
    assert launches.get("k_own_multi", 0) > 0, launches               # some groups did share launches
    if fuse_ep:
        assert launches.get("k_evalpart_multi", 0) > 0, launches


def single(flags):
    """adversarial batches through guber_eval_batch (host pointers: stage, copies, kernels, copies back) on one engine"""
    eng, orc = ga.Engine(cache_size=4096, max_batch=4096, flags=flags), support.Oracle(cache_size=1 << 20)
    for k, b in enumerate(streams.adversarial_batches(21, 8, 1500, greg_fn=support.gregorian)):
        want, got = orc.eval(b), eng.eval(b)
        support.assert_results_equal(got, want, f"batch {k}")
        assert got.counters() == want.counters(), k
    assert eng.size() == orc.size()
    eng.close()


def dev_gen(hb, full):
    """a generation for guber_front_eval_dev: every column a "device" array (full: burst / created_at / is_owner too)"""
    n = hb.n
    cols = dict(key_bytes=np.ascontiguousarray(np.concatenate([hb.key_bytes, np.zeros(16, np.uint8)])), key_off=np.ascontiguousarray(hb.key_off.view(np.int32)),
                hits=np.ascontiguousarray(hb.hits), limit=np.ascontiguousarray(hb.limit), duration=np.ascontiguousarray(hb.duration),
                algorithm=np.ascontiguousarray(hb.algorithm), behavior=np.ascontiguousarray(hb.behavior.view(np.int32)),
                burst=np.ascontiguousarray(hb.burst) if (full and hb.burst is not None) else None,
                created_at=np.ascontiguousarray(hb.created_at) if (full and hb.created_at is not None) else None,
                is_owner=np.ascontiguousarray(hb.is_owner) if (full and hb.is_owner is not None) else None)
    p = {k: (v.ctypes.data if v is not None else None) for k, v in cols.items()}
    r = dict(status=np.full(n, 99, np.uint8), err=np.full(n, 99, np.uint8), limit=np.full(n, -7, np.int64), remaining=np.full(n, -7, np.int64), reset_time=np.full(n, -7, np.int64))
    b = ga.GuberBatch(n, 0, p["key_bytes"], p["key_off"], p["hits"], p["limit"], p["duration"], p["burst"], p["created_at"],
                      p["algorithm"], p["behavior"], p["is_owner"], None, None, hb.now_ms)
    res = ga.GuberResult(r["status"].ctypes.data, r["limit"].ctypes.data, r["remaining"].ctypes.data, r["reset_time"].ctypes.data, r["err"].ctypes.data, 0, 0, 0, 0, 0)
    return b, res, cols, r


def front(n_engines, n_streams, fuse_ep, max_batch=4096):
    """guber_front_eval_dev: ONE stream of requests in arrival order over ONE key space -> k_fr_count / k_fr_scatter hand every request to
    the engine the placement names (hot keys placed individually) -> the engines' shares through the fused launches (shares larger than an
    engine's max_batch in pieces) -> k_fr_out brings the answers back in arrival order.  Twelve generations in two calls: fixed-width keys
    (they travel with their requests) and ragged ones (they stay where they are), every request column or only the mandatory ones, an
    empty generation, a generation of three requests, uniform keys.  Every generation equals ONE oracle fed the generations in order."""
    K, G = 9000, 8192
    tab = streams.key_table(K)
    place = ga.Placement(n_engines)
    place.observe_keys(*streams.keys_for_ids(tab, streams.ZipfSampler(K, seed=77).draw(1 << 15)))
    place.rebalance(0.125, True)
    engs = []                                                       # engines of one stream are neighbours (the dispatcher groups neighbours)
    for j in range(n_engines):
        sj = j * n_streams // n_engines
        first = next((q for q in range(j) if q * n_streams // n_engines == sj), None)
        engs.append(ga.Engine(cache_size=1 << 16, max_batch=max_batch, stream=None if first is None else engs[first].stream_handle()))
    for e in engs:
        e.profile(True)
    fr = ga.Front(engs, place, max_n=G, depth=4)
    orc = support.Oracle(cache_size=1 << 20)
    zs = streams.ZipfSampler(K, seed=31)
    rng = np.random.default_rng(8)
    adv = streams.adversarial_batches(44, 6, 3000, greg_fn=support.gregorian)
    gens = []
    for g in range(12):
        now = streams.NOW0 + g * 700
        if g in (3, 7, 10):
            hb = next(adv)                                          # ragged keys, every column, every behaviour but host-computed calendars
            hb.behavior[:] = hb.behavior & ~np.uint32(4)
            hb.duration[:] = np.where(hb.duration < 8, 50, hb.duration)
            hb.greg_expire[:] = 0
            hb.greg_duration[:] = 0
            gens.append((hb, True))
            continue
        n = [G, 5000, 0, G, 3, G, 7777, 6000, G, 1, G, 300][g]
        ids = rng.integers(0, K, n) if g in (5, 8) else zs.draw(n)
        hb = streams.bench_batch(tab, ids, now, algorithm=g % 2, limit=30, duration=4000)
        if n:
            hb.algorithm[:] = (np.arange(n) // 97 + g) % 2           # both algorithms inside one generation
        gens.append((hb, False))
    done = 0
    for lo, hi in ((0, 5), (5, 12)):
        part = [dev_gen(hb, full) for hb, full in gens[lo:hi]]
        N = hi - lo
        got_done = fr.eval_dev((ga.GuberBatch * N)(*[x[0] for x in part]), (ga.GuberResult * N)(*[x[1] for x in part]), N)
        fr.synchronize()
        assert got_done == N, (got_done, N)
        for k, (hb, full) in enumerate(gens[lo:hi]):
            want = orc.eval(hb)
            got = ga.HostResult(hb.n)
            for name in ("status", "limit", "remaining", "reset_time", "err"):
                getattr(got, name)[:hb.n] = part[k][3][name]
            if hb.n:
                support.assert_results_equal(got, want, f"generation {lo + k}")
        done += N
    launches = {}
    for e in engs:
        for k, v in e.profile_read().items():
            launches[k] = launches.get(k, 0) + v[0]
    print("launches", {k: v for k, v in launches.items() if v}, "front", fr.stats())
    assert launches.get("k_fr_count", 0) == launches.get("k_fr_scatter", 0) == launches.get("k_fr_out", 0) == launches.get("k_fr_scan", 0) == 11, launches   # (the empty generation launches nothing)
    # tables of ONE stream and generations of at most 131 072 requests: all tables in one pair of launches (launch_group_mem); the tests that
    # are about the owner-partitioned pipeline under the front switch that off (GUBER_FRONT_ONE_PAIR_MAX=0, a laboratory knob)
    one_pair = n_streams == 1 and n_engines > 1 and os.environ.get("GUBER_FRONT_ONE_PAIR_MAX") != "0"
    if one_pair:
        assert launches.get("k_front_multi", 0) >= 8 and launches.get("k_own_multi", 0) == 0, launches
    elif n_engines > 1 and max_batch >= 4096:
        assert launches.get("k_own_multi", 0) > 0, launches
        if fuse_ep and max_batch >= 4096:
            assert launches.get("k_evalpart_multi", 0) > 0, launches
    assert fr.stats()["generations"] == 12
    assert sum(e.size() for e in engs) == orc.size(), ([e.size() for e in engs], orc.size())
    sizes = [e.size() for e in engs]
    assert min(sizes) > 0, sizes                                    # every engine holds a part of the key space
    fr.close()
    for e in engs:
        e.close()
    place.close()


def bench_sequence():
    """bench.py's own sequence in small, on the CPU build of the engine under AddressSanitizer: twelve engines over three
    streams with the product's placement; the residency pass (hits = 0, a batch per engine at a time through guber_eval_batch_dev); a
    pre-split stretch (guber_eval_batches_routed_dev: the stream split by the placement, a shard flushing whenever B of its requests wait);
    then the routed arrangement (guber_front_eval_dev over generations of four batches, warm-up and timed calls) — every "device" buffer a
    host allocation, so a kernel or a copy that reads or writes outside one is a report; answers equal ONE oracle fed the same requests"""
    S, K, B = 12, 60_000, 4096
    tab = streams.key_table(K)
    place = ga.Placement(S)
    place.observe_keys(*streams.keys_for_ids(tab, streams.ZipfSampler(K, seed=990_001).draw(1 << 16)))
    place.rebalance(0.125, True)
    sown, _ = place.route_keys(*streams.keys_for_ids(tab, np.arange(K)))
    heads = {}
    engs = []
    for j in range(S):
        sj = j * 3 // S
        engs.append(ga.Engine(cache_size=2 * int((sown == j).sum()) + 8 * B + 1024, max_batch=B, stream=heads[sj].stream_handle() if sj in heads else None))
        heads.setdefault(sj, engs[-1])
    orc = support.Oracle(cache_size=4 * K)
    now = streams.NOW0
    keep = []
    for j in range(S):                                               # residency pass
        loc = np.nonzero(sown == j)[0]
        for lo in range(0, len(loc), B):
            hb = streams.bench_batch(tab, loc[lo:lo + B], now, hits=0)
            b, res, cols, rd = dev_batch(hb)
            engs[j].eval_dev(b, res)
            engs[j].synchronize()
            orc.eval(hb)
    assert sum(e.size() for e in engs) == orc.size() == K
    zs = streams.ZipfSampler(K, seed=1234)
    ids = zs.draw(40 * B)
    # pre-split: per shard the stream's requests in order, flushed B at a time, in flush order
    which, hbs, cb, cr = [], [], [], []
    per = [np.nonzero(sown[ids] == j)[0] for j in range(S)]
    flush = sorted((int(per[j][(q + 1) * B - 1]), j, q) for j in range(S) for q in range(len(per[j]) // B))
    for step, (_, j, q) in enumerate(flush):
        hb = streams.bench_batch(tab, ids[per[j][q * B:(q + 1) * B]], now + 1 + step)
        b, res, cols, rd = dev_batch(hb)
        keep.append((cols, rd)); which.append(j); hbs.append(hb); cb.append(b); cr.append(res)
    N = len(which)
    ga.Engine.eval_routed_dev(engs, (C.c_uint32 * N)(*which), (ga.GuberBatch * N)(*cb), (ga.GuberResult * N)(*cr), N)
    for s in range(N):
        want = orc.eval(hbs[s])
        got = ga.HostResult(hbs[s].n)
        for name in ("status", "limit", "remaining", "reset_time", "err"):
            getattr(got, name)[:] = keep[s][1][name]
        support.assert_results_equal(got, want, f"pre-split batch {s} of table {which[s]}")
    # routed: generations of four batches, a warm-up call and a timed one
    fr = ga.Front(engs, place, max_n=4 * B, depth=4)
    ids2 = zs.draw(12 * 4 * B).reshape(12, 4 * B)
    gens = [streams.bench_batch(tab, ids2[g], now + 1000 + 4 * g) for g in range(12)]
    for lo, hi in ((0, 4), (4, 12)):
        part = [dev_gen(hb, False) for hb in gens[lo:hi]]
        n = hi - lo
        assert fr.eval_dev((ga.GuberBatch * n)(*[x[0] for x in part]), (ga.GuberResult * n)(*[x[1] for x in part]), n) == n
        fr.synchronize()
        for k, hb in enumerate(gens[lo:hi]):
            want = orc.eval(hb)
            got = ga.HostResult(hb.n)
            for name in ("status", "limit", "remaining", "reset_time", "err"):
                getattr(got, name)[:] = part[k][3][name]
            support.assert_results_equal(got, want, f"generation {lo + k}")
    assert fr.stats()["forced_flushes"] == 0 and sum(e.stats()["retries"] for e in engs) == 0
    assert sum(e.size() for e in engs) == orc.size()
    fr.close()
    for e in engs:
        e.close()
    place.close()


def front_lru(workers):
    """tests/test_gpu_front.py test_the_front_over_binding_caches_is_the_references_worker_pool on the CPU build of the engine"""
    cs, nkeys, G = 2000, 2600, 4096
    place = ga.Placement(workers)
    e0 = ga.Engine(cache_size=cs // workers, max_batch=4096)
    engs = [e0] + [ga.Engine(cache_size=cs // workers, max_batch=4096, stream=e0.stream_handle()) for _ in range(workers - 1)]
    fr = ga.Front(engs, place, max_n=G, depth=4)
    orc = support.Oracle(cache_size=cs, workers=workers)
    for step, hb in enumerate(streams.length_changing_batches(29, 6, nkeys, 3000, "reset+greg", support.gregorian)):
        hb.greg_expire[:] = 0
        want = orc.eval(hb)
        b, res, cols, rd = dev_gen(hb, True)
        assert fr.eval_dev((ga.GuberBatch * 1)(b), (ga.GuberResult * 1)(res), 1) == 1
        fr.synchronize()
        got = ga.HostResult(hb.n)
        for name in ("status", "limit", "remaining", "reset_time", "err"):
            getattr(got, name)[:hb.n] = rd[name]
        support.assert_results_equal(got, want, f"generation {step}")
        assert sum(e.size() for e in engs) == orc.size(), (step, [e.size() for e in engs], orc.size())
    assert sum(e.stats()["unexpired_evictions"] for e in engs) == orc.counters()[3]
    fr.close()
    for e in engs:
        e.close()
    place.close()


def front_global():
    """the rule's global_engine: requests that carry Behavior_GLOBAL go to the device's GLOBAL engine whatever their key hashes to (as the
    pool's routing does: a device keeps the keys of GLOBAL requests in an engine of its own), the others by the placement.  Keys are either
    always GLOBAL or never in this stream, so ONE oracle sees the same sequence per key; the GLOBAL engine ends up with exactly those keys."""
    K, G = 3000, 4096
    tab = streams.key_table(K)
    place = ga.Placement(2)
    e0 = ga.Engine(cache_size=1 << 15, max_batch=4096)
    engs = [e0, ga.Engine(cache_size=1 << 15, max_batch=4096, stream=e0.stream_handle()), ga.Engine(cache_size=1 << 15, max_batch=4096, stream=e0.stream_handle())]
    fr = ga.Front(engs, place, max_n=G, depth=4, global_engine=2)
    orc = support.Oracle(cache_size=1 << 20)
    zs = streams.ZipfSampler(K, seed=41)
    for g in range(5):
        ids = zs.draw(G)
        hb = streams.bench_batch(tab, ids, streams.NOW0 + g * 500, algorithm=g % 2, limit=40, duration=5000)
        hb.behavior[:] = np.where(ids % 3 == 0, 2, 0).astype(np.uint32)
        b, res, cols, rd = dev_gen(hb, False)
        assert fr.eval_dev((ga.GuberBatch * 1)(b), (ga.GuberResult * 1)(res), 1) == 1
        fr.synchronize()
        want = orc.eval(hb)
        got = ga.HostResult(hb.n)
        for name in ("status", "limit", "remaining", "reset_time", "err"):
            getattr(got, name)[:] = rd[name]
        support.assert_results_equal(got, want, f"generation {g}")
    seen = np.unique(np.concatenate([streams.ZipfSampler(K, seed=41).draw(5 * G)]))
    assert engs[2].size() == int((seen % 3 == 0).sum()), (engs[2].size(), int((seen % 3 == 0).sum()))
    assert engs[0].size() + engs[1].size() == int((seen % 3 != 0).sum()) and min(engs[0].size(), engs[1].size()) > 0
    fr.close()
    for e in engs:
        e.close()
    place.close()


CASES = {
    "front_global": front_global,
    "front_lru3": lambda: front_lru(3),
    "bench_sequence": bench_sequence,
    "front4": lambda: front(4, 1, os.environ.get("GUBER_FUSE_EP") == "1"),
    "front6x2": lambda: front(6, 2, os.environ.get("GUBER_FUSE_EP") == "1"),
    "front6x3_pieces": lambda: front(6, 3, os.environ.get("GUBER_FUSE_EP") == "1", max_batch=1024),
    "front1": lambda: front(1, 1, os.environ.get("GUBER_FUSE_EP") == "1"),
    "routed4": lambda: routed(4, os.environ.get("GUBER_FUSE_EP") == "1"),
    "routed6": lambda: routed(6, os.environ.get("GUBER_FUSE_EP") == "1"),
    "routed_threads": lambda: routed_with_a_second_thread(os.environ.get("GUBER_FUSE_EP") == "1"),
    "routed_lru": lambda: routed_lru(os.environ.get("GUBER_FUSE_EP") == "1"),
    "single_default": lambda: single(0),
    "single_part": lambda: single(ga.FLAG_TEST_FORCE_PART),
}

if __name__ == "__main__":
    CASES[sys.argv[1]]()
    print("ENGINESIM CASE OK", sys.argv[1])
