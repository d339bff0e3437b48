"""The batch pipelines' KERNEL SOURCE (gubernator_amd/csrc/guber_kernels.h: k_front / k_eval2, guber_kernels_part.h: k_part /
k_own / k_eval3) compiled for the host against tests/hostsim/fakehip and run one workgroup at a time, every device thread a
cooperative fiber, against the oracle — on a machine without a GPU.  A de-risking aid for the kernels' logic (grouping, rank
bases, flags, probing and inserting, the serial walk of heterogeneous segments); the parity claim itself is the `-m gpu` suite."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import streams
from support import GREGORIAN, ROOT, GuberBatch, GuberResult, HostBatch, HostResult, Oracle, assert_results_equal, gregorian

HS = os.path.join(ROOT, "tests", "hostsim")


# the kernels as the product builds them (64-byte messages, 32-byte records that leave out what the request says), with the owner count
# (128 or 256 per batch: Work::pmode) following the traffic and pinned either way
@pytest.fixture(scope="module", params=[("libdevsim.so", 0), ("libdevsim.so", 7), ("libdevsim.so", 8)],
                ids=lambda p: p[0][3:-3] + (f"-owners{1 << p[1]}" if p[1] else ""))
def lib(request):
    subprocess.run(["make", "-s", "-C", HS, "devsim_lib"], check=True)
    request_param, owner_bits = request.param
    L = C.CDLL(os.path.join(HS, request_param))
    L.ds_create.restype = C.c_void_p
    L.ds_create.argtypes = [C.c_uint64, C.c_uint32, C.c_int]
    L.ds_destroy.argtypes = [C.c_void_p]
    L.ds_eval.argtypes = [C.c_void_p, C.POINTER(GuberBatch), C.POINTER(GuberResult), C.c_int, C.c_int]
    L.ds_counters.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    L.ds_chaos.argtypes = [C.c_uint32]
    L.ds_create_bounded.restype = C.c_void_p
    L.ds_create_bounded.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_uint64]
    L.ds_lru_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    L.ds_part_forms.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_ulonglong)]
    L.ds_pin_owner_bits.argtypes = [C.c_void_p, C.c_uint32]
    L.ds_owner_bits.argtypes = [C.c_void_p]
    L.ds_owner_bits.restype = C.c_uint32
    L.ds_owner_hold.argtypes = [C.c_void_p, C.c_uint32]
    L.ds_owner_hold.restype = C.c_uint32
    L.owner_bits = owner_bits
    L.product_form = owner_bits == 0
    return L


def only_where_the_form_matters(lib, pipeline=1):
    """the other builds differ in k_part / k_own / k_eval3 only: what does not run them, or is about something else, runs once"""
    if not lib.product_form and pipeline != 1:
        pytest.skip("the two-launch pipeline is the same in every build")


class Sim:
    def __init__(self, lib, slots=4096, max_batch=4096, weak=0, pipeline=1, cache_size=0):
        self.lib, self.pipeline = lib, pipeline
        self.h = lib.ds_create_bounded(slots, max_batch, weak, cache_size)
        if lib.owner_bits:
            lib.ds_pin_owner_bits(self.h, lib.owner_bits)

    def owner_bits(self):
        return self.lib.ds_owner_bits(self.h)

    def lru_stats(self):
        out = (C.c_ulonglong * 6)()
        self.lib.ds_lru_stats(self.h, out)
        return dict(zip(("admits", "applied", "rebuilds", "cuts", "passes", "unexpired_evictions"), out))

    def eval(self, batch, careful=0):
        res = HostResult(batch.n)
        rc = self.lib.ds_eval(self.h, C.byref(batch.c), C.byref(res.c), self.pipeline, careful)
        assert rc == 0, rc
        return res

    def part_forms(self, n):
        """(key, tile) groups of the last owner-partitioned batch of n requests, 32-byte records among them, messages with the tile's shape"""
        out = (C.c_ulonglong * 3)()
        self.lib.ds_part_forms(self.h, n, out)
        return tuple(out)

    def counters(self):
        out = (C.c_longlong * 6)()
        self.lib.ds_counters(self.h, out)
        return tuple(out)

    def close(self):
        self.lib.ds_destroy(self.h)


def sub_batch(b, idx):
    """the requests `idx` of batch b as a batch of their own (the engine's retry round does the same by index list)"""
    keys = [bytes(b.key_bytes[b.key_off[i]:b.key_off[i + 1]]) for i in idx]
    def col(a, dt=None):
        return None if a is None else np.asarray(a)[idx]
    return HostBatch(keys, col(b.hits), col(b.limit), col(b.duration), b.now_ms, burst=col(b.burst), created_at=col(b.created_at),
                     algorithm=col(b.algorithm), behavior=col(b.behavior), is_owner=col(b.is_owner),
                     greg_expire=col(b.greg_expire), greg_duration=col(b.greg_duration))


@pytest.mark.parametrize("pipeline", [1, 0])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_adversarial_streams_through_the_kernel_source(lib, pipeline, seed):
    """every branch of algorithms.go, duplicate-heavy keys, mixed request shapes: element-wise equal to the oracle, counters too"""
    only_where_the_form_matters(lib, pipeline)
    if not lib.product_form and seed != 1:
        pytest.skip("the other builds and modes run one seed")
    sim, orc = Sim(lib, pipeline=pipeline), Oracle()
    for k, b in enumerate(streams.adversarial_batches(seed, 10, 1500, greg_fn=gregorian)):
        want, got = orc.eval(b), sim.eval(b)
        assert_results_equal(got, want, f"seed {seed} batch {k}")
    o, hi, mi, sz, retries, _ = sim.counters()
    assert retries == 0
    co = orc.counters()
    assert (o, hi, mi) == (co[0], co[1], co[2]) and sz == orc.size()
    sim.close()


def test_zipf_batches_with_hot_keys_spanning_every_tile(lib):
    """5000-request Zipf batches over 3000 keys (hot keys in all 20 tiles: rank bases across tiles and chunks), token and leaky,
    the clock stepping so that buckets leak, expire and renew"""
    sim, orc = Sim(lib, slots=16384, max_batch=5000), Oracle()
    table = streams.key_table(3000)
    z = streams.ZipfSampler(3000, seed=7)
    now = streams.NOW0
    for k in range(8):
        ids = z.draw(5000)
        b = streams.bench_batch(table, ids, now, algorithm=k & 1, limit=40, duration=3000)
        assert_results_equal(sim.eval(b), orc.eval(b), f"batch {k}")
        groups, short_recs, _ = sim.part_forms(5000)
        # The 32-byte record serves every key whose bucket
        # holds what the request says (limit, duration, burst — or no burst) and every new key: all of batch 0 (new) and batch 1
        # (leaky requests meeting token buckets of the same limit), most of the leaky batches after that; a token request that
        # meets a leaky bucket (burst 40 stored, none asked for) gets the 64-byte form — both forms in one batch from batch 2 on
        assert groups > 2000
        assert short_recs == groups if k < 2 else 0 < short_recs < groups, (k, groups, short_recs)
        now += [1, 700, 1, 3500, 2, 900, 1, 1][k]
    assert sim.counters()[:3] == orc.counters()[:3]
    sim.close()


def test_owner_rounds_split_when_a_round_does_not_fit(lib):
    """a table so small that all keys share ONE owner: 2304 distinct keys in a batch (> OW_MCAP messages, > OW_KCAP keys) force the
    owner's round to split by further bits of the home position, repeatedly"""
    sim, orc = Sim(lib, slots=1 << 20, max_batch=2304), Oracle()
    # keys whose home positions all fall into owner 0: pick them by hash
    from support import oracle_lib
    ol = oracle_lib()
    keys, i = [], 0
    while len(keys) < 700:
        k = b"own_%d" % i
        i += 1
        if ((ol.oracle_xxhash64(k, len(k), 0) >> 7) & ((1 << 20) - 1)) >> 12 == 0:
            keys.append(k)
    rng = np.random.default_rng(5)
    now = streams.NOW0
    for rnd in range(3):
        ids = np.concatenate([np.arange(700), rng.integers(0, 700, 1604)])
        rng.shuffle(ids)
        b = HostBatch([keys[j] for j in ids], 1, 5, 60000, now + rnd)
        assert_results_equal(sim.eval(b), orc.eval(b), f"round {rnd}")
    sim.close()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("nkeys", [513, 600, 768])
def test_more_keys_of_one_owner_than_its_lds_table_has_cells(lib, nkeys):
    """513 .. 768 DISTINCT keys of one owner in one batch: few enough messages for one round (<= OW_MCAP), more keys than the round's
    LDS hash table has cells (OW_HT = 512).  The insert loop used to probe the full table for ever (a hang of k_own — found on the GPU
    with 128 owners per batch and uniform keys, reachable with 256 by keys chosen to share an owner); it is bounded now and the round
    splits"""
    only_where_the_form_matters(lib, 1 if lib.product_form or lib.owner_bits else 0)
    from support import oracle_lib
    ol = oracle_lib()
    keys, i = [], 0
    while len(keys) < nkeys:
        k = b"own_%d" % i
        i += 1
        if ((ol.oracle_xxhash64(k, len(k), 0) >> 7) & ((1 << 20) - 1)) >> 12 == 0:
            keys.append(k)
    sim, orc = Sim(lib, slots=1 << 20, max_batch=2304), Oracle()
    for rnd in range(2):
        b = HostBatch(keys, 1, 5, 60000, streams.NOW0 + rnd)
        assert_results_equal(sim.eval(b), orc.eval(b), f"round {rnd}")
    # ... and keys that come back in other tiles of the batch (several groups per key, rank bases across tiles) while the owner's
    # round has to split: 450 of the keys, 750 requests in 3 tiles
    rng = np.random.default_rng(nkeys)
    for rnd in range(2):
        ids = np.concatenate([np.arange(450), rng.integers(0, 450, 300)])
        rng.shuffle(ids)
        b = HostBatch([keys[j] for j in ids], 1, 5, 60000, streams.NOW0 + 10 + rnd, algorithm=(ids % 2).astype(np.uint8))
        assert_results_equal(sim.eval(b), orc.eval(b), f"split round, round {rnd}")
    sim.close()


def test_the_owner_count_follows_the_traffic(lib):
    """Work::pmode: a batch starts with 128 owners; when 8 or more of them split a round for their number of keys (uniform keys:
    512 per owner), the batch's k_eval3 moves the next 255 batches to 256 owners, then probes 128 again — answers equal to the
    oracle all the way (guber_kernels_part.h "HOW MANY OWNERS")"""
    if lib.owner_bits or not lib.product_form:
        pytest.skip("the traffic-following mode of the product's form")
    sim, orc = Sim(lib, slots=1 << 20, max_batch=8192), Oracle(cache_size=1 << 20)
    assert sim.owner_bits() == 7
    table = streams.key_table(20_000)
    now = streams.NOW0
    # Zipf batches: nobody splits, the count stays at 128
    z = streams.ZipfSampler(20_000, seed=11)
    for k in range(2):
        b = streams.bench_batch(table, z.draw(6_000), now + k, limit=50, duration=60_000)
        assert_results_equal(sim.eval(b), orc.eval(b), f"zipf {k}")
        assert sim.owner_bits() == 7
    # 400 distinct keys for each of 8 of the 128 owners (more than a round's 320): those eight split; the NEXT batch runs with 256 owners
    from support import oracle_lib
    ol = oracle_lib()
    crowd, i = [], 0
    per = [0] * 8
    while min(per) < 400:
        k = b"crowd_%d" % i
        i += 1
        o = ((ol.oracle_xxhash64(k, len(k), 0) >> 7) & ((1 << 20) - 1)) >> 13
        if o < 8 and per[o] < 400:
            per[o] += 1
            crowd.append(k)
    for rnd in range(2):
        b = HostBatch(crowd, 1, 50, 60_000, now + 10 + rnd)
        assert_results_equal(sim.eval(b), orc.eval(b), f"crowded owners {rnd}")
        assert sim.owner_bits() == 8
    # ... for 255 batches, then 128 again (the test shortens the wait: 254 left -> 4 left)
    assert lib.ds_owner_hold(sim.h, 0) == 254
    lib.ds_owner_hold(sim.h, 4)
    small = streams.bench_batch(table, z.draw(300), now + 12, limit=50, duration=60_000)
    for k in range(4):
        assert_results_equal(sim.eval(small), orc.eval(small), f"hold {k}")
        assert sim.owner_bits() == (8 if k < 3 else 7), k
    assert_results_equal(sim.eval(small), orc.eval(small), "128 owners again")
    assert sim.owner_bits() == 7
    sim.close()


@pytest.mark.parametrize("pipeline", [1, 0])
def test_hash_collisions_are_reported_and_resolved_by_the_careful_round(lib, pipeline):
    """6 significant hash bits: distinct keys share hashes; the pipelines answer RETRY for those and nothing else, and the careful
    round of the two-launch pipeline (what the engine re-runs them through) gives the oracle's answers"""
    only_where_the_form_matters(lib, pipeline)
    sim, fix, orc = Sim(lib, weak=1, pipeline=pipeline), None, Oracle()
    rng = np.random.default_rng(11)
    keys = [b"col_%d" % i for i in range(300)]
    now = streams.NOW0
    retried = 0
    for rnd in range(4):
        ids = rng.integers(0, 300, 900)
        b = HostBatch([keys[j] for j in ids], 1, 50, 60000, now + rnd)
        # the engine's way (eval_batch_host_locked): the batch, then careful rounds over what was answered RETRY until nothing is;
        # the oracle sees the requests in the order they were really applied
        pending, rounds = np.arange(b.n), 0
        while len(pending):
            sb = sub_batch(b, pending)
            if rounds == 0:
                got = sim.eval(sb)
            else:
                simc = Sim.__new__(Sim); simc.lib, simc.pipeline, simc.h = lib, 0, sim.h
                got = simc.eval(sb, careful=1)
            err = np.asarray(got.err)
            ok = np.nonzero(err != 5)[0]
            assert len(ok), "a round must answer something"
            assert_results_equal(_take(got, ok), orc.eval(sub_batch(sb, ok)), f"round {rnd}.{rounds}")
            pending = pending[np.nonzero(err == 5)[0]]
            retried += len(pending)
            rounds += 1
            assert rounds < 64
    assert retried > 0
    sim.close()


def _take(res, idx):
    out = HostResult(len(idx))
    for name in ("status", "limit", "remaining", "reset_time", "err"):
        getattr(out, name)[:] = np.asarray(getattr(res, name))[idx]
    return out


def test_scheduling_order_does_not_matter(lib):
    """the same stream with the fibers scheduled in alternating order and yielding inside atomics: same answers (catches a missing
    barrier between an LDS write and its readers)"""
    lib.ds_chaos(1)
    try:
        sim, orc = Sim(lib), Oracle()
        for k, b in enumerate(streams.adversarial_batches(9, 6, 1200, greg_fn=gregorian)):
            assert_results_equal(sim.eval(b), orc.eval(b), f"batch {k}")
        sim.close()
    finally:
        lib.ds_chaos(0)


def test_the_routing_kernels_agree_with_the_placement(lib):
    """guber_kernels_route.h (k_route_count + k_route_dest: what guber_stage_route launches) compiled for the host: every request's
    shard = guber_placement_route_keys (workers.go:153-155, 180-184 generalised: slot table + hot-key list), GLOBAL requests go to
    the GLOBAL engine, ranks follow the arrival order, the shares' sizes add up — before and after a rebalance that pins hot
    keys; also with the workgroups in a shuffled order (the tile scan runs in whichever workgroup finishes last)."""
    only_where_the_form_matters(lib, 0)
    import gubernator_amd as ga
    lib.ds_route.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64,
                             C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)
    n_plain, glob = 5, 5
    pl = ga.Placement(n_plain)
    for rnd, n in enumerate([1, 255, 256, 257, 3000, 9000, 700]):
        if rnd == 4:
            ids = rng.zipf(1.1, 100_000) % 20_000
            seen = HostBatch([f"k_{int(i)}" + "z" * int(i % 13) for i in ids], 1, 1, 1, 0)
            pl.observe_keys(seen.key_bytes, seen.key_off)
            pl.rebalance(0.05, move_slots=True)
            assert pl.n_hot() > 0
        lib.ds_chaos(rnd if rnd % 2 else 0)
        rule = pl.export(global_engine=glob)
        ids = rng.zipf(1.1, n) % 20_000
        hb = HostBatch([f"k_{int(i)}" + "z" * int(i % 13) for i in ids], 1, 1, 1, 0, behavior=np.where(rng.random(n) < 0.15, 2, 0))
        dest, counts = np.zeros(n, np.uint32), np.zeros(16, np.uint32)
        rc = lib.ds_route(hb.key_bytes.ctypes.data, hb.key_off.ctypes.data, hb.behavior.ctypes.data, n, n_plain + 1, 1024, rule.n_shards, rule.per, rule.step,
                          rule.inv_step, rule.inv_sub, rule.table, rule.ex_cells, rule.ex_n, rule.ex_hash, rule.ex_shard, glob, dest.ctypes.data, counts.ctypes.data)
        assert rc == 0, rc
        shard, _ = pl.route_keys(hb.key_bytes, hb.key_off)
        shard = shard.astype(np.uint32).copy()
        shard[(hb.behavior & 2) != 0] = glob
        want_counts = np.bincount(shard, minlength=16).astype(np.uint32)
        assert np.array_equal(counts, want_counts), (rnd, counts, want_counts)
        assert np.array_equal(dest >> 24, shard), rnd
        for e in range(n_plain + 1):
            assert np.array_equal(dest[shard == e] & 0xffffff, np.arange(want_counts[e], dtype=np.uint32)), (rnd, e)   # arrival order
    lib.ds_chaos(0)
    pl.close()


@pytest.mark.parametrize("pipeline", [1, 0])
@pytest.mark.parametrize("pattern", ["cyclic", "random", "zipf", "expiring"])
def test_bounded_cache_evicts_in_the_reference_order(lib, pipeline, pattern):
    """lrucache.go:88-149 through the eviction pre-pass (guber_kernels_lru.h) + an unchanged pipeline: 2 600 keys over a cache of
    2 000, batches of 1 500 in which evicted keys come back in the same and in the next batch — every answer, the size after every
    batch and the count of unexpired evictions equal the bounded-LRU oracle.  cyclic = the classic worst case (every access of an
    exact LRU misses), expiring = short durations with the clock moving (expired items still hold their place in the list)."""
    only_where_the_form_matters(lib, 0)          # (the pre-pass does not depend on the owner count)
    cs, nkeys, bsz = 2000, 2600, 1500
    sim, orc = Sim(lib, slots=1 << 15, max_batch=4096, pipeline=pipeline, cache_size=cs), Oracle(cache_size=cs)
    rng = np.random.default_rng(11)
    z = streams.ZipfSampler(nkeys, seed=3)
    now, pos = streams.NOW0, 0
    for step in range(16):
        if pattern == "cyclic":
            ids = (pos + np.arange(bsz)) % nkeys
            pos += bsz
        elif pattern == "zipf":
            ids = z.draw(bsz)
        else:
            ids = rng.integers(0, nkeys, bsz)
        dur = 1500 if pattern == "expiring" else 3_600_000
        b = HostBatch([f"lru_{int(i)}" for i in ids], 1, 1000, dur, now, algorithm=(ids & 1).astype(np.uint8))
        want, got = orc.eval(b), sim.eval(b)
        assert_results_equal(got, want, f"{pattern} step {step}")
        assert sim.counters()[3] == orc.size() <= cs, step
        now += 1000
    st = sim.lru_stats()
    assert st["unexpired_evictions"] == orc.counters()[3] and st["applied"] >= 1, st
    assert sim.counters()[:3] == orc.counters()[:3]
    sim.close()


@pytest.mark.parametrize("pipeline", [1, 0])
def test_a_new_key_whose_requests_all_fail_in_the_algorithm_is_no_insert(lib, pipeline):
    """algorithms.go: a request with DURATION_IS_GREGORIAN and a duration that is no interval constant (interval.go:93,107,125,148) fails
    in tokenBucketNewItem / leakyBucketNewItem BEFORE c.Add: for a key that is not resident it is a GetItem miss and nothing else — no
    insert, nobody leaves the list for it (lrucache.go:98-100).  A key with a failing request first and a good one later is inserted
    by the good one.  Under a binding cache every answer, the size after every batch and the unexpired evictions equal the oracle's."""
    only_where_the_form_matters(lib, 0)
    cs, nkeys, bsz = 2000, 2600, 1500
    sim, orc = Sim(lib, slots=1 << 15, max_batch=4096, pipeline=pipeline, cache_size=cs), Oracle(cache_size=cs)
    rng = np.random.default_rng(17)
    now = streams.NOW0
    for step in range(12):
        ids = rng.integers(0, nkeys, bsz)
        keys = [f"lru_{int(i)}" for i in ids]
        beh = np.zeros(bsz, np.uint32); dur = np.full(bsz, 3_600_000, np.int64)
        bad = rng.choice(bsz, 120, replace=False)
        for q, i in enumerate(bad):                              # keys nobody else asks for: all their requests fail
            keys[i] = f"never_{step}_{q % 40}"                   # (some of them three times in the batch)
            beh[i] = GREGORIAN; dur[i] = 3 if q % 2 else 99      # weeks (not supported) / no interval constant
        late = rng.choice(np.setdiff1d(np.arange(bsz), bad), 60, replace=False)
        for q, i in enumerate(sorted(late)):                     # a failing request first, a good one for the same key later in the batch
            keys[i] = f"late_{step}_{q // 2}"
            if q % 2 == 0:
                beh[i] = GREGORIAN; dur[i] = 77
        ge, gd = np.zeros(bsz, np.int64), np.zeros(bsz, np.int64)
        for i in np.nonzero(beh & GREGORIAN)[0]:                 # (as the host layer precomputes them: a negative greg_duration carries the error)
            ge[i], gd[i] = gregorian(now, int(dur[i]))
        b = HostBatch(keys, 1, 1000, dur, now, algorithm=(ids & 1).astype(np.uint8), behavior=beh, greg_expire=ge, greg_duration=gd)
        want, got = orc.eval(b), sim.eval(b)
        assert_results_equal(got, want, f"step {step}")
        assert (got.err[bad] != 0).all()
        assert sim.counters()[3] == orc.size() <= cs, (step, sim.counters()[3], orc.size())
        now += 1000
    st = sim.lru_stats()
    assert st["unexpired_evictions"] == orc.counters()[3] and st["applied"] >= 1, (st, orc.counters())
    sim.close()


@pytest.mark.parametrize("pipeline,kinds", [(1, "reset+greg"), (0, "greg")])
def test_requests_that_change_the_lists_length_are_evaluated_on_their_own(lib, pipeline, kinds):
    """Under a binding cache the eviction pre-pass models every access as "the key is at the front now".  Two kinds of request are not
    that: a TOKEN_BUCKET RESET_REMAINING of a key in the cache removes the item and inserts nothing (algorithms.go:78-90), and a resident
    key's first request that fails before c.Add (an invalid Gregorian constant) is no arrival at the front if the key had been pushed out
    before it, or had expired (lrucache.go:111-128).  The pre-pass reports the first such request (LRU_SPLIT) and the host evaluates the
    requests before it, it alone, and the rest as batches of their own: every answer, the size after every batch and the unexpired
    evictions equal the bounded-LRU oracle's (the second kind was round 5's documented divergence; the first was found
    writing this test — both fail without the split)."""
    only_where_the_form_matters(lib, 0)
    cs, nkeys, bsz = 2000, 2600, 1500
    sim, orc = Sim(lib, slots=1 << 15, max_batch=4096, pipeline=pipeline, cache_size=cs), Oracle(cache_size=cs)
    for step, b in enumerate(streams.length_changing_batches(23, 4, nkeys, bsz, kinds, gregorian)):
        want, got = orc.eval(b), sim.eval(b)
        assert_results_equal(got, want, f"{kinds} step {step}")
        assert sim.counters()[3] == orc.size() <= cs, (step, sim.counters()[3], orc.size())
    st = sim.lru_stats()
    assert st["unexpired_evictions"] == orc.counters()[3] and st["applied"] >= 1 and st["cuts"] >= 1, (st, orc.counters())
    sim.close()


def test_batches_larger_than_the_cache_are_cut(lib):
    """a cache of 300 items under batches of 1 000 requests over 500 keys: the batch is evaluated in pieces of cache_size requests,
    a key evicted by request i is a new item for request j > i of the same batch (lrucache.go:98-100)"""
    only_where_the_form_matters(lib, 0)
    cs = 300
    sim, orc = Sim(lib, slots=1 << 14, max_batch=4096, cache_size=cs), Oracle(cache_size=cs)
    rng = np.random.default_rng(5)
    now = streams.NOW0
    for step in range(6):
        ids = rng.integers(0, 500, 1000)
        b = HostBatch([f"cut_{int(i)}" for i in ids], 1, 50, 3_600_000, now)
        assert_results_equal(sim.eval(b), orc.eval(b), f"step {step}")
        assert sim.counters()[3] == orc.size() == cs
        now += 10
    st = sim.lru_stats()
    assert st["cuts"] >= 5 and st["unexpired_evictions"] == orc.counters()[3], st
    sim.close()


# ---- GUBER_FUSE_EP: batch b's k_eval3 and batch b + 1's k_part of the same tables in ONE launch (k_evalpart_multi) --------------------
def ep_lib(lib):
    if not lib.product_form:
        pytest.skip("the fused launch is built from the product's form")
    lib.ds_fuse_ep.argtypes = [C.c_void_p, C.c_int]
    lib.ds_block_order.argtypes = [C.c_uint32]
    lib.ds_owner_slot.argtypes = [C.c_void_p, C.c_uint32]
    lib.ds_owner_slot.restype = C.c_uint32
    lib.ds_eval_stream_ep.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(GuberBatch), C.POINTER(GuberResult), C.c_uint32]
    return lib


def run_ep_stream(lib, sims, rounds):
    """rounds[r][j] = the batch of table j in round r: through ds_eval_stream_ep (k_part_multi, k_own_multi, then per round ONE
    k_evalpart_multi + k_own_multi, the last k_eval3_multi); the results, same shape"""
    nh, nr = len(sims), len(rounds)
    bs, rs = (GuberBatch * (nh * nr))(), (GuberResult * (nh * nr))()
    res = [[HostResult(b.n) for b in rnd] for rnd in rounds]
    for r in range(nr):
        for j in range(nh):
            bs[r * nh + j], rs[r * nh + j] = rounds[r][j].c, res[r][j].c
    hs = (C.c_void_p * nh)(*[s.h for s in sims])
    rc = lib.ds_eval_stream_ep(hs, nh, bs, rs, nr)
    assert rc == 0, rc
    return res


@pytest.mark.parametrize("order,chaos", [(0, 0), (1, 0), (2, 1)], ids=["eval3_first", "part_first", "shuffled_chaos"])
def test_eval3_and_the_next_batchs_part_in_one_launch(lib, order, chaos):
    """three tables, six rounds of adversarial batches each (every branch of algorithms.go, hot keys, mixed shapes, 1 .. 7 tiles):
    whichever half of the fused launch runs first — all of k_eval3(b) before k_part(b + 1), the reverse, or the workgroups
    shuffled — every batch of every table equals its oracle, counters too (the halves share nothing: packed words per batch parity,
    the owner count from the parity's slot)"""
    L = ep_lib(lib)
    sims, orcs = [Sim(L, slots=8192, max_batch=2048) for _ in range(3)], [Oracle() for _ in range(3)]
    for s in sims:
        L.ds_fuse_ep(s.h, 1)
    gens = [streams.adversarial_batches(40 + j, 6, [1700, 600, 300][j], greg_fn=gregorian) for j in range(3)]
    rounds = [[next(g) for g in gens] for _ in range(6)]
    L.ds_block_order(order)
    L.ds_chaos(chaos)
    try:
        res = run_ep_stream(L, sims, rounds)
    finally:
        L.ds_block_order(0)
        L.ds_chaos(0)
    for j in range(3):
        for r in range(6):
            assert_results_equal(res[r][j], orcs[j].eval(rounds[r][j]), f"table {j} round {r}")
        o, hi, mi, sz, retries, _ = sims[j].counters()
        co = orcs[j].counters()
        assert retries == 0 and (o, hi, mi) == (co[0], co[1], co[2]) and sz == orcs[j].size()
    for s in sims:
        s.close()


def test_a_fuse_ep_table_launched_batch_by_batch(lib):
    """what guber_engine.hip does for such an engine when nothing can be fused (a lone batch, maintenance in between): the three
    launches one after the other, the packed words and the owner count still per batch parity"""
    L = ep_lib(lib)
    sim, orc = Sim(L), Oracle()
    L.ds_fuse_ep(sim.h, 1)
    for k, b in enumerate(streams.adversarial_batches(3, 8, 1500, greg_fn=gregorian)):
        assert_results_equal(sim.eval(b), orc.eval(b), f"batch {k}")
    sim.close()


def test_the_owner_count_of_a_fused_stream_follows_one_batch_later(lib):
    """a batch whose k_eval3 shares a launch with the next batch's k_part cannot change THAT batch's owner count any more: it leaves
    the decision in its parity's slot, for the batch after the next (pm_bits_of) — crowded owners in batch 1 move batch 3 to 256
    owners, batch 2 still splits its rounds at 128; the answers equal the oracle's all the way, in either order of the halves"""
    L = ep_lib(lib)
    from support import oracle_lib
    ol = oracle_lib()
    crowd, i, per = [], 0, [0] * 8
    while min(per) < 400:
        k = b"crowd_%d" % i
        i += 1
        o = ((ol.oracle_xxhash64(k, len(k), 0) >> 7) & ((1 << 20) - 1)) >> 13
        if o < 8 and per[o] < 400:
            per[o] += 1
            crowd.append(k)
    table = streams.key_table(20_000)
    z = streams.ZipfSampler(20_000, seed=11)
    now = streams.NOW0
    for order in (0, 1):
        sim, orc = Sim(L, slots=1 << 20, max_batch=8192), Oracle(cache_size=1 << 20)
        L.ds_fuse_ep(sim.h, 1)
        rounds = [[streams.bench_batch(table, z.draw(3_000), now, limit=50, duration=60_000)],
                  [HostBatch(crowd, 1, 50, 60_000, now + 1)],
                  [HostBatch(crowd, 1, 50, 60_000, now + 2)],
                  [HostBatch(crowd, 1, 50, 60_000, now + 3)],
                  [streams.bench_batch(table, z.draw(3_000), now + 4, limit=50, duration=60_000)]]
        L.ds_block_order(order)
        try:
            res = run_ep_stream(L, [sim], rounds[:2])          # batches 0 and 1: the second one crowds eight owners
            assert (L.ds_owner_slot(sim.h, 0), L.ds_owner_slot(sim.h, 1)) == (7, 8)      # batch 2 (parity 0) stays at 128, batch 3 gets 256
            res += run_ep_stream(L, [sim], rounds[2:])
        finally:
            L.ds_block_order(0)
        assert (L.ds_owner_slot(sim.h, 0), L.ds_owner_slot(sim.h, 1)) == (8, 8)
        for r, rnd in enumerate(rounds):
            assert_results_equal(res[r][0], orc.eval(rnd[0]), f"order {order} batch {r}")
        sim.close()
