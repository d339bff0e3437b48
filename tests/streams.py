"""Seeded synthetic request streams shared by the CPU-side logic tests, the GPU parity tests and
bench.py.  Everything is numpy; keys are built as byte matrices so 10M-key streams stay cheap."""
import numpy as np

from support import HostBatch

NOW0 = 1_700_000_000_000


def key_table(n_keys, prefix=b"bench_k", digits=8):
    """Fixed-width keys  prefix + zero-padded id  -> (n_keys, L) uint8 matrix.  'bench_k00000042' is
    name 'bench' + '_' + unique_key 'k00000042' (client.go:39-41)."""
    ids = np.arange(n_keys, dtype=np.int64)
    L = len(prefix) + digits
    m = np.empty((n_keys, L), np.uint8)
    m[:, :len(prefix)] = np.frombuffer(prefix, np.uint8)
    for d in range(digits):
        m[:, L - 1 - d] = (ids // 10 ** d) % 10 + ord("0")
    return m


def keys_for_ids(table, ids):
    sel = table[ids]
    n, L = sel.shape
    key_bytes = np.ascontiguousarray(sel).reshape(-1)
    key_bytes = np.concatenate([key_bytes, np.zeros(8, np.uint8)])
    key_off = (np.arange(n + 1, dtype=np.uint32) * L).astype(np.uint32)
    return key_bytes, key_off


class ZipfSampler:
    """Exact inverse-CDF Zipf(s) over ranks 1..n_keys, rank -> key id through a fixed permutation
    (s = 1.1, stream seed 1234, permutation seed 99)."""

    def __init__(self, n_keys, s=1.1, seed=1234, perm_seed=99):
        w = 1.0 / np.power(np.arange(1, n_keys + 1, dtype=np.float64), s)
        self.cdf = np.cumsum(w)
        self.cdf /= self.cdf[-1]
        self.perm = np.random.default_rng(perm_seed).permutation(n_keys)
        self.rng = np.random.default_rng(seed)

    def draw(self, n):
        u = self.rng.random(n)
        ranks = np.searchsorted(self.cdf, u, side="left")
        return self.perm[np.minimum(ranks, len(self.perm) - 1)]


def bench_batch(table, ids, now_ms, algorithm=0, hits=1, limit=100, duration=60_000):
    """The BASELINE config-2/3 request shape: hits 1, limit 100, duration 60 s, behaviour 0,
    created_at = now (NULL array), burst 0."""
    kb, ko = keys_for_ids(table, ids)
    return HostBatch((kb, ko), hits, limit, duration, now_ms, algorithm=algorithm)


def adversarial_batches(seed, n_batches, batch_size, n_keys=97, greg_fn=None):
    """Random batches that try to hit every branch of algorithms.go: duplicate-heavy key draws, both
    algorithms (and invalid ones), RESET / DRAIN / GREGORIAN bits, zero / negative / huge hits,
    limit and duration changes, burst changes, per-request created_at, clocks that cross expiry."""
    rng = np.random.default_rng(seed)
    now = NOW0
    for _ in range(n_batches):
        n = int(batch_size if rng.random() < 0.7 else rng.integers(1, batch_size + 1))
        hot = rng.random() < 0.5
        if hot:   # a few very hot keys -> long same-key runs inside the batch
            ids = np.where(rng.random(n) < 0.8, rng.integers(0, 3, n), rng.integers(0, n_keys, n))
        else:
            ids = rng.integers(0, n_keys, n)
        keys = [b"adv_" + str(int(i)).encode() + (b"x" * int(i % 7) * 11) for i in ids]
        uniform = rng.random() < 0.5   # whole batch shares one parameter set (closed-form path)
        def pick(choices, p=None):
            if uniform:
                return np.full(n, rng.choice(choices, p=p))
            return rng.choice(choices, size=n, p=p)
        hits = pick([0, 1, 1, 1, 2, 3, 7, -1, -3, 1000, 2 ** 62, -(2 ** 62)])
        limit = pick([0, 1, 2, 5, 10, 10, 10, 100, 2000, -5, 2 ** 62])
        duration = pick([0, 1, 5, 50, 50, 50, 1000, 30000, 60000, 2 ** 40])
        burst = pick([0, 0, 0, 5, 20, -1])
        algorithm = pick([0, 0, 1, 1, 7], p=[0.4, 0.1, 0.4, 0.08, 0.02]).astype(np.uint8)
        beh = np.zeros(n, np.uint32)
        for bit, p in [(8, 0.03), (32, 0.2), (4, 0.05), (2, 0.05), (1, 0.05)]:
            if uniform:
                if rng.random() < p:
                    beh |= bit
            else:
                beh |= np.where(rng.random(n) < p, bit, 0).astype(np.uint32)
        if rng.random() < 0.5:
            created = np.full(n, now)
        elif uniform:
            created = np.full(n, now + int(rng.integers(-100, 100)))
        else:
            created = now + rng.integers(-70, 70, n)
        is_owner = (rng.random(n) < 0.8).astype(np.uint8)
        ge = np.zeros(n, np.int64)
        gd = np.zeros(n, np.int64)
        greg = (beh & 4) != 0
        if greg.any():
            # gregorian requests carry an interval selector in `duration`
            sel = rng.integers(0, 8, n)
            duration = np.where(greg, sel, duration)
            for i in np.nonzero(greg)[0]:
                ge[i], gd[i] = greg_fn(now, int(duration[i]))
        yield HostBatch(keys, hits, limit, duration, now, burst=burst, created_at=created, algorithm=algorithm,
                        behavior=beh, is_owner=is_owner, greg_expire=ge, greg_duration=gd)
        now += int(rng.choice([0, 1, 1, 3, 10, 60, 1000, 61000]))


def length_changing_batches(seed, steps, nkeys, bsz, kinds, greg_fn):
    """batches over `nkeys` keys in which some requests change the LENGTH of the reference's list instead of moving a key to its front
    (guber_kernels_lru.h "ISOLATED"): kinds has "reset" — TOKEN_BUCKET requests with RESET_REMAINING on keys of the population (resident or
    not, as a key's first, middle and last request of the batch) — and / or "greg" — requests with DURATION_IS_GREGORIAN and no interval
    constant as the FIRST request of keys of the population, many of them among the oldest of a binding cache; a third of the keys have a
    duration short enough to have expired when they are asked for again.  -> HostBatch per step (calendar values precomputed)"""
    rng = np.random.default_rng(seed)
    now = NOW0
    for step in range(steps):
        ids = rng.integers(0, nkeys, bsz)
        beh = np.zeros(bsz, np.uint32)
        dur = np.where(ids % 3 == 0, 1500, 3_600_000).astype(np.int64)
        algo = (ids & 1).astype(np.uint8)
        if "reset" in kinds:
            for i in rng.choice(bsz, 40, replace=False):
                beh[i] |= 8                                          # Behavior_RESET_REMAINING
                algo[i] = 0 if rng.random() < 0.8 else algo[i]       # (a leaky request's RESET refills the bucket and removes nothing)
            hot = int(ids[0])                                        # one key with several requests around its RESETs
            for i in rng.choice(bsz, 12, replace=False):
                ids[i] = hot; algo[i] = 0; dur[i] = 3_600_000 if hot % 3 else 1500
        if "greg" in kinds:
            seen = set()
            for i in range(bsz):                                     # the FIRST request of some keys of the population cannot insert
                k = int(ids[i])
                if k not in seen and rng.random() < 0.06:
                    beh[i] |= 4; dur[i] = 77 if rng.random() < 0.5 else 3
                seen.add(k)
        ge, gd = np.zeros(bsz, np.int64), np.zeros(bsz, np.int64)
        for i in np.nonzero(beh & 4)[0]:
            ge[i], gd[i] = greg_fn(now, int(dur[i]))
        yield HostBatch([f"lru_{int(i)}" for i in ids], 1, 1000, dur, now, algorithm=algo, behavior=beh, greg_expire=ge, greg_duration=gd)
        now += 1000
