"""CPU differential tests of the KERNEL LOGIC (gubernator_amd/csrc/guber_algo.h: apply / skip /
eval_uniform_rank, XXH64, FNV) compiled for the host by tests/hostsim — against the oracle.  These
de-risk the HIP path on a machine with no GPU; the parity tests proper are the -m gpu ones."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import xxhash

import scenarios
import streams
import support
from support import HostBatch, HostResult, GuberBatch, GuberResult, GuberItem, Oracle

HS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim")
_lib = None


def hostsim_lib():
    global _lib
    if _lib is None:
        so = os.path.join(HS_DIR, "libhostsim.so")
        srcs = [os.path.join(HS_DIR, "hostsim.cpp"), os.path.join(support.ROOT, "gubernator_amd/csrc/guber_algo.h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.run(["g++", "-O2", "-g", "-fPIC", "-std=c++17", "-fwrapv", "-ffp-contract=off", "-shared",
                            "-o", so, srcs[0]], check=True)
        lib = C.CDLL(so)
        lib.hs_create.restype = C.c_void_p
        lib.hs_destroy.argtypes = [C.c_void_p]
        lib.hs_path_counts.argtypes = [C.c_void_p, C.c_void_p]
        lib.hs_path_counts.restype = None
        lib.hs_eval_batch.argtypes = [C.c_void_p, C.POINTER(GuberBatch), C.POINTER(GuberResult), C.c_int]
        lib.hs_add_item.argtypes = [C.c_void_p, C.POINTER(GuberItem), C.POINTER(C.c_int)]
        lib.hs_get_item.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_int64, C.POINTER(GuberItem),
                                    C.POINTER(C.c_int)]
        lib.hs_size.restype = C.c_int64
        lib.hs_size.argtypes = [C.c_void_p]
        lib.hs_xxhash64.restype = C.c_uint64
        lib.hs_xxhash64.argtypes = [C.c_char_p, C.c_uint32, C.c_uint64]
        lib.hs_fnv1_64.restype = C.c_uint64
        lib.hs_fnv1_64.argtypes = [C.c_char_p, C.c_uint32]
        lib.hs_fnv1a_64.restype = C.c_uint64
        lib.hs_fnv1a_64.argtypes = [C.c_char_p, C.c_uint32]
        _lib = lib
    return _lib


class HostSim:
    def __init__(self, mode=0):
        self.lib = hostsim_lib()
        self.h = self.lib.hs_create()
        self.mode = mode

    def close(self):
        if self.h:
            self.lib.hs_destroy(self.h)
            self.h = None

    def eval(self, batch):
        res = HostResult(batch.n)
        self.lib.hs_eval_batch(self.h, C.byref(batch.c), C.byref(res.c), self.mode)
        return res

    def path_counts(self):
        """(multi-request segments on the parallel closed-form path, walked serially)"""
        out = (C.c_uint64 * 2)()
        self.lib.hs_path_counts(self.h, out)
        return out[0], out[1]

    def add_item(self, item, now_ms=0):
        ex = C.c_int(0)
        self.lib.hs_add_item(self.h, C.byref(item), C.byref(ex))
        return bool(ex.value)

    def get_item(self, key, now_ms):
        kb = key if isinstance(key, bytes) else key.encode()
        out, found = GuberItem(), C.c_int(0)
        self.lib.hs_get_item(self.h, kb, len(kb), now_ms, C.byref(out), C.byref(found))
        return support.item_dict(out, kb) if found.value else None

    def size(self):
        return self.lib.hs_size(self.h)

    def each(self):
        raise NotImplementedError


def test_device_hashes_match_independent_implementations():
    lib = hostsim_lib()
    olib = support.oracle_lib()
    rng = np.random.default_rng(3)
    for n in list(range(0, 100)) + [255, 256, 1000]:
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert lib.hs_xxhash64(b, n, 0) == xxhash.xxh64(b, seed=0).intdigest()
        assert lib.hs_xxhash64(b, n, 77) == xxhash.xxh64(b, seed=77).intdigest()
        assert lib.hs_fnv1_64(b, n) == olib.oracle_fnv1_64(b, n)
        assert lib.hs_fnv1a_64(b, n) == olib.oracle_fnv1a_64(b, n)


def test_golden_vectors_through_kernel_logic():
    assert scenarios.run_functional(lambda: HostSim()) >= 75
    for case in scenarios.load("store_vectors.json")["cases"]:
        case.pop("expect_size", None)

    # store vectors (no Each on the host sim)
    import json
    n = 0
    orig = scenarios.load

    def load_no_size(name):
        d = orig(name)
        if name == "store_vectors.json":
            for c in d["cases"]:
                c.pop("expect_size", None)
        return d
    scenarios.load = load_no_size
    try:
        n = scenarios.run_store(lambda: HostSim())
    finally:
        scenarios.load = orig
    assert n == 5


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_adversarial_streams_closed_form_vs_oracle(seed):
    o, h = Oracle(cache_size=1 << 20), HostSim(mode=0)
    total = 0
    for bi, b in enumerate(streams.adversarial_batches(seed, 120, 400, greg_fn=support.gregorian)):
        want, got = o.eval(b), h.eval(b)
        support.assert_results_equal(got, want, f"seed {seed} batch {bi}")
        assert got.counters()[:3] == want.counters()[:3], f"counters seed {seed} batch {bi}"
        assert h.size() == o.size(), f"size seed {seed} batch {bi}"
        total += b.n
    assert total > 20000


def test_hot_key_uniform_runs_long():
    """One key hit thousands of times in a batch (the Zipf head): ranks far beyond the bucket's
    remaining, with and without DRAIN, both algorithms, fractional leaky remainders."""
    now = streams.NOW0
    for algo in (0, 1):
        for beh in (0, 32):
            for hits, limit in [(1, 100), (3, 100), (7, 1000), (1, 5000), (5, 5)]:
                o, h = Oracle(cache_size=1 << 16), HostSim(mode=0)
                for step in range(4):
                    n = 6000
                    keys = [b"hot_key"] * n
                    b = HostBatch(keys, hits, limit, 60_000, now + step * 1700, algorithm=algo, behavior=beh)
                    support.assert_results_equal(h.eval(b), o.eval(b), f"algo {algo} beh {beh} hits {hits} step {step}")


def test_zipf_bench_stream_small():
    tab = streams.key_table(50_000)
    z = streams.ZipfSampler(50_000)
    for algo in (0, 1):
        o, h = Oracle(cache_size=1 << 20), HostSim(mode=0)
        for bi in range(6):
            b = streams.bench_batch(tab, z.draw(8192), streams.NOW0 + bi * 20_000, algorithm=algo)
            support.assert_results_equal(h.eval(b), o.eval(b), f"algo {algo} batch {bi}")


def test_random_uniform_runs_vs_oracle():
    """Many short scenarios: one key, a random pre-state built by 1-3 random requests, then a run of
    n identical requests whose rank-wise answers must match the sequential oracle."""
    rng = np.random.default_rng(99)
    now = streams.NOW0
    for trial in range(1500):
        o, h = Oracle(cache_size=1 << 12), HostSim(mode=0)
        t = now
        for phase in range(int(rng.integers(1, 4))):
            n = int(rng.choice([1, 2, 3, 17, 64, 300]))
            hits = int(rng.choice([0, 1, 2, 3, 5, 9, -1, 50]))
            limit = int(rng.choice([0, 1, 7, 10, 100, 250]))
            duration = int(rng.choice([0, 3, 100, 1000, 60000]))
            algo = int(rng.choice([0, 1]))
            beh = int(rng.choice([0, 0, 0, 32, 8, 40]))
            burst = int(rng.choice([0, 0, 15, 300]))
            created = t + int(rng.choice([0, 0, -5, 5, -2000]))
            b = HostBatch([b"solo_key"] * n, hits, limit, duration, t, burst=burst, created_at=created,
                          algorithm=algo, behavior=beh)
            support.assert_results_equal(h.eval(b), o.eval(b), f"trial {trial} phase {phase}")
            assert h.size() == o.size()
            t += int(rng.choice([0, 1, 2, 40, 150, 1200, 70000]))
        o.close(); h.close()


def test_created_at_only_variation_on_hot_token_keys():
    """Requests of one key that differ only in created_at (RPCs stamped a ms apart): parallel path for live
    token buckets, serial otherwise — answers must equal the oracle in every case."""
    rng = np.random.default_rng(17)
    now = streams.NOW0
    for algo in (0, 1):
        o, h = Oracle(cache_size=1 << 12), HostSim(mode=0)
        for step in range(6):
            n = 900
            keys = [b"hotk"] * n
            created = now + step * 400 + np.sort(rng.integers(0, 3, n))
            beh = 8 if step == 4 else 0
            dur = 60_000 if step != 3 else 30_000        # step 3 changes the duration -> created_at matters
            b = HostBatch(keys, 1, 5000, dur, now + step * 400 + 2, created_at=created, algorithm=algo, behavior=beh)
            support.assert_results_equal(h.eval(b), o.eval(b), f"algo {algo} step {step}")


def test_created_at_only_variation_on_hot_leaky_keys():
    """Aggregated RPC payloads stamp a hot LEAKY key's requests a few ms apart.  The run stays on the parallel path
    only while no request leaks and none pulls the expiration before the batch clock (leaky_created_harmless);
    every mix — harmless, one leaking member in the middle, created_at far in the past / future, tiny durations,
    fresh and expired buckets — must equal the oracle bit for bit."""
    rng = np.random.default_rng(23)
    now = streams.NOW0
    cases, paths = 0, []
    for dur, limit in ((60_000, 100), (1000, 100_000), (3, 10), (60_000, 7)):
        o, h = Oracle(cache_size=1 << 12), HostSim(mode=0)
        t = now
        for step in range(14):
            n = int(rng.integers(2, 700))
            kind = step % 7
            base = t - int(rng.integers(0, 3))
            created = base + np.sort(rng.integers(0, 4, n))                 # harmless: within a few ms
            if kind == 2:
                created[n // 2] += int(rng.choice([dur, 10 * dur, 700, 5]))     # one member leaks (or not, by rate)
            elif kind == 3:
                created[rng.integers(0, n)] = t - 10 * dur - 5                   # far in the past: expiry guard
            elif kind == 4:
                created = t + rng.integers(-2000, 2000, n)                       # unsorted, wide
            elif kind == 5:
                created[0] += 3 * dur                                            # the claimer itself is the odd one
            hits = 1 if kind != 6 else 0
            b = HostBatch([b"hot_leaky"] * n, hits, limit, dur, t, created_at=created, algorithm=1,
                          behavior=32 if step == 9 else 0)
            support.assert_results_equal(h.eval(b), o.eval(b), f"dur {dur} limit {limit} step {step} kind {kind}")
            assert h.size() == o.size()
            t += int(rng.choice([1, 3, 50, dur // 2 + 1, 2 * dur]))
            cases += 1
        par, ser = h.path_counts()
        paths.append((par, ser))
        o.close(); h.close()
    assert cases == 56
    assert sum(p for p, _ in paths) >= 12 and sum(s_ for _, s_ in paths) >= 12, paths   # both paths exercised


def test_extreme_value_runs_vs_oracle():
    """The closed forms (skip / eval_uniform_rank) under Go's wrap-around and float->int rules: int64 extremes and
    negatives for hits / limit / duration / burst / created_at, items pre-loaded with extreme Remaining (token int64,
    leaky float64 incl. fractions, negatives, 1e300), then runs of identical requests — rank by rank equal to the
    sequential oracle."""
    rng = np.random.default_rng(2024)
    now = streams.NOW0
    I64 = [0, 1, -1, 2, 3, 7, 100, 2**31, 2**53, 2**53 + 1, 2**62, -(2**62), 2**63 - 1, -(2**63), 2**63 - 2, -(2**63) + 1]
    F64 = [0.0, 0.5, 1.0, 1.5, -1.0, -0.25, 99.999, 2.0**53, 2.0**53 + 2, 9.3e18, -9.3e18, 1e300, -1e300, 3.0, 10.0]
    for trial in range(2500):
        o, h = Oracle(cache_size=1 << 12), HostSim(mode=0)
        t = now + int(rng.integers(0, 10_000))
        algo0 = int(rng.integers(0, 2))
        if rng.random() < 0.7:                       # pre-load an item with an extreme state
            it = dict(limit=int(rng.choice(I64)), duration=int(rng.choice([1000, 60_000, 0, -5, 2**62])), remaining=int(rng.choice(I64)),
                      remaining_f=float(rng.choice(F64)), stamp=t - int(rng.choice([0, 1, 999, 10**9])), burst=int(rng.choice(I64[:8] + [2**62])),
                      expire_at=t + int(rng.choice([0, 1, 60_000, -1, 2**62])))
            for be in (o, h):
                be.add_item(support.make_item("xk", algo0, **it), t)
        for phase in range(int(rng.integers(1, 4))):
            n = int(rng.choice([1, 2, 3, 5, 40, 200]))
            hits = int(rng.choice(I64 + [1, 1, 1, 2, 5]))
            limit = int(rng.choice(I64 + [10, 100]))
            duration = int(rng.choice([0, 1, 3, 1000, 60_000, -1, -(2**62), 2**62, 2**63 - 1]))
            algo = int(rng.choice([0, 1]))
            beh = int(rng.choice([0, 0, 32, 8, 40]))
            burst = int(rng.choice([0, 0, 15, -3, 2**62, 2**63 - 1]))
            created = int(rng.choice([t, t, t - 5, t + 5, 0, -1, 2**62, -(2**62), t - 10**9]))
            b = HostBatch([b"xk"] * n, hits, limit, duration, t, burst=burst, created_at=created, algorithm=algo, behavior=beh)
            support.assert_results_equal(h.eval(b), o.eval(b), f"trial {trial} phase {phase} algo {algo} hits {hits} limit {limit} dur {duration} "
                                                            f"burst {burst} created {created} beh {beh}")
            assert h.size() == o.size()
            a, b_ = o.get_item("xk", t), h.get_item("xk", t)
            if a is None or b_ is None:
                assert a is None and b_ is None, (trial, phase)
            else:
                for f in ("algorithm", "status", "limit", "duration", "remaining", "stamp", "burst", "expire_at"):
                    assert a[f] == b_[f], (trial, phase, f, a, b_)
                assert a["remaining_f"] == b_["remaining_f"] or (a["remaining_f"] != a["remaining_f"] and b_["remaining_f"] != b_["remaining_f"]), (trial, a, b_)
            t += int(rng.choice([0, 1, 40, 1200, 70_000]))
        o.close(); h.close()


def test_closed_forms_equal_apply_skip_on_random_states():
    """token_fast / leaky_fast (guber_algo.h: the counter-walk closed forms the kernels try first) against
    eval_uniform_rank (apply + skip) on random live buckets and requests, ranks 0 .. 2^20: responses, event flags and
    the bucket after the request must be identical wherever a closed form accepts the case."""
    lib = hostsim_lib()
    lib.hs_fuzz_closed_forms.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p]
    lib.hs_fuzz_closed_forms.restype = None
    tok = leaky = 0
    for seed in range(1, 9):
        out = (C.c_uint64 * 3)()
        lib.hs_fuzz_closed_forms(seed, 40_000, out)
        assert out[2] == 0, f"seed {seed}: {out[2]} mismatches"
        tok += out[0]; leaky += out[1]
    assert tok > 200_000 and leaky > 100_000, (tok, leaky)


def _gregorian_batches(seed, n_batches=40, greg_fn=None, now0=None):
    """(batch with host-precomputed calendar values, the same batch without them).  greg_fn(now_ms, d) -> (expire, duration):
    default = the oracle's (UTC)"""
    rng = np.random.default_rng(seed)
    now = streams.NOW0 if now0 is None else now0
    greg_fn = greg_fn or support.gregorian
    for _ in range(n_batches):
        n = int(rng.integers(1, 300))
        keys = [b"greg_%d" % int(i) for i in rng.integers(0, 12, n)]
        hits = rng.choice([0, 1, 1, 2, 7], n)
        limit = rng.choice([5, 10, 100], n)
        sel = rng.choice([0, 1, 2, 3, 4, 5, 9], n, p=[0.25, 0.2, 0.2, 0.05, 0.15, 0.1, 0.05])     # interval selector carried in `duration`
        algo = rng.integers(0, 2, n).astype(np.uint8)
        beh = np.where(rng.random(n) < 0.9, 4, 0).astype(np.uint32) | np.where(rng.random(n) < 0.1, 32, 0).astype(np.uint32)
        dur = np.where(beh & 4, sel, 60_000)
        ge, gd = np.zeros(n, np.int64), np.zeros(n, np.int64)
        for i in range(n):
            if beh[i] & 4:
                ge[i], gd[i] = greg_fn(now, int(dur[i]))
        yield (HostBatch(keys, hits, limit, dur, now, algorithm=algo, behavior=beh, greg_expire=ge, greg_duration=gd),
               HostBatch(keys, hits, limit, dur, now, algorithm=algo, behavior=beh))
        now += int(rng.choice([1, 1000, 61_000, 3_600_000, 86_400_000 * 20]))


def test_gregorian_intervals_computed_from_the_batch_clock():
    """DURATION_IS_GREGORIAN without host-precomputed values: the kernel logic derives the calendar interval (interval.go:84-148,
    UTC) from the batch clock itself; answers equal the oracle fed with the host-computed values, errors included."""
    o, h = Oracle(cache_size=1 << 12), HostSim(mode=0)
    for bi, (with_vals, without) in enumerate(_gregorian_batches(5)):
        support.assert_results_equal(h.eval(without), o.eval(with_vals), f"batch {bi}")
