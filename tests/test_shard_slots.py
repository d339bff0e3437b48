"""guber_placement_* (csrc/placement.cpp): placement of a GPU's keys on its logical shards — hash slots whose initial table IS the
reference's worker rule (workers.go:153-155,180-184), plus individually placed hot key hashes fitted to observed traffic."""
import numpy as np
import xxhash

import gubernator_amd as ga
import streams
import support


def _keys(nk, ids=None):
    tab = streams.key_table(nk)
    return streams.keys_for_ids(tab, np.arange(nk) if ids is None else ids)


def test_initial_placement_is_the_reference_worker_rule():
    o = support.oracle_lib()
    for workers in (1, 2, 3, 5, 8, 12, 13):
        pl = ga.Placement(workers)
        rng = np.random.default_rng(workers)
        for h in [int(x) for x in rng.integers(0, 2 ** 63, 400, dtype=np.int64)] + [0, 1, 2 ** 64 - 1, 2 ** 63, 2 ** 63 - 1]:
            # (hash63 / hashRingStep can reach `workers` when workers does not divide 2^63 — the reference would index past its slice;
            #  the engine gives those last few hashes to the last worker)
            assert pl.shard(h) == min(o.oracle_worker_index_for_hash63(workers, h >> 1), workers - 1), (workers, h)
        k = b"shardkey_77"
        assert pl.shard(xxhash.xxh64(k, seed=0).intdigest()) == min(o.oracle_worker_index_for_hash63(workers, xxhash.xxh64(k, seed=0).intdigest() >> 1), workers - 1)
        pl.close()


def test_placement_balances_a_zipf_stream_and_isolates_the_hot_keys():
    nk, S = 200_000, 12
    pl = ga.Placement(S)
    observed = streams.ZipfSampler(nk, s=1.1, seed=7).draw(1 << 20)
    pl.observe_keys(*_keys(nk, observed))
    moves = pl.rebalance(0.125, True)
    assert pl.n_hot() >= 3 and len(moves) == pl.n_hot()
    sown, hashes = pl.route_keys(*_keys(nk))
    assert sown.min() >= 0 and sown.max() < S
    later = streams.ZipfSampler(nk, s=1.1, seed=1234).draw(1 << 21)           # the stream that is measured later
    share = np.bincount(sown[later], minlength=S) / len(later)
    hot = np.bincount(later).max() / len(later)
    assert share.max() <= max(hot, 1.0 / S) * 1.10, share                     # no shard above the hottest key's own share
    rest = np.sort(share)[:-1]
    assert rest.max() / rest.min() < 1.25, share                              # the others are even
    plain = ga.Placement(S)                                                    # the untouched worker rule: the hot key's shard carries its 1/S on top
    psown, _ = plain.route_keys(*_keys(nk))
    pshare = np.bincount(psown[later], minlength=S) / len(later)
    assert pshare.max() > share.max() * 1.2, (pshare, share)
    top = int(np.bincount(later).argmax())
    assert int(hashes[top]) in {m[0] for m in moves}                           # the hottest key is placed individually
    # deterministic: the same observations give the same placement
    pl2 = ga.Placement(S)
    pl2.observe_keys(*_keys(nk, observed))
    pl2.rebalance(0.125, True)
    assert np.array_equal(pl2.route_keys(*_keys(nk))[0], sown)
    pl.close(); pl2.close(); plain.close()


def test_online_pass_moves_only_hot_keys_and_reports_them():
    nk, S = 50_000, 8
    pl = ga.Placement(S)
    before, hashes = pl.route_keys(*_keys(nk))
    before = before.copy()
    ids = np.concatenate([np.full(60_000, 4242), np.random.default_rng(3).integers(0, nk, 140_000)])
    pl.observe_keys(*_keys(nk, ids))
    moves = pl.rebalance(0.125, False)                                         # slots stay; hot keys get the least loaded shard
    after, _ = pl.route_keys(*_keys(nk))
    changed = np.nonzero(after != before)[0]
    assert set(changed.tolist()) <= {4242}
    assert all(m[0] == int(hashes[4242]) for m in moves) and len(moves) <= 1
    assert pl.n_hot() == 1
    pl.close()


def test_individual_places_age_out_when_the_hot_set_drifts():
    """online passes (the slot table stays): a key that stopped being heavy three passes ago follows its slot again — reported as a move
    back, so that its bucket can be migrated — and the 64 individual places never fill up with yesterday's hot keys (placement.cpp
    plan_locked).  An idle pass (too few observations) forgets nothing."""
    nk, S = 50_000, 8
    pl = ga.Placement(S)
    slot_shard, hashes = pl.route_keys(*_keys(nk))
    slot_shard = slot_shard.copy()
    rng = np.random.default_rng(5)

    def window(hot_ids, n=200_000):
        ids = np.concatenate([np.repeat(np.asarray(hot_ids), 12_000), rng.integers(0, nk, n)])
        pl.observe_keys(*_keys(nk, ids))
        return pl.rebalance(0.125, False)

    first = [101, 202, 303]
    moves = window(first)
    assert pl.n_hot() == 3
    pinned = {int(hashes[k]): pl.shard(int(hashes[k])) for k in first}
    assert all(m[0] in pinned and m[2] == pinned[m[0]] for m in moves)
    # an idle pass: a handful of requests says nothing about what is hot
    pl.observe_keys(*_keys(nk, rng.integers(0, nk, 100)))
    assert pl.rebalance(0.125, False) == [] and pl.n_hot() == 3
    # the hot set drifts: other keys are heavy now; the old ones cool for three passes, then go
    back = []
    for rnd in range(4):
        second = [4000 + rnd * 7 + j for j in range(3)]                          # (a new trio every pass: drift)
        mv = window(second)
        back += [m for m in mv if m[0] in pinned]
        if rnd < 2:
            assert all(pl.shard(h) == s for h, s in pinned.items()), rnd          # not yet: two cold passes are not three
    assert all(pl.shard(int(hashes[k])) == slot_shard[k] for k in first)         # they follow their slots again
    assert {m[0] for m in back} == {h for h, s in pinned.items() if s != slot_shard[[int(x) for x in hashes].index(h)]}
    assert all(m[1] == pinned[m[0]] and m[2] == pl.shard(m[0]) for m in back)    # reported from the individual place back to the slot's shard
    assert pl.n_hot() <= 3 * 3                                                    # only the recent trios hold places (each ages out in turn)
    # many more drifting rounds never exhaust the 64 places
    for rnd in range(40):
        window([10_000 + rnd * 5 + j for j in range(4)])
        assert pl.n_hot() <= 4 * 4, (rnd, pl.n_hot())
    latest = [10_000 + 39 * 5 + j for j in range(4)]
    assert all(pl.shard(int(hashes[k])) is not None for k in latest)
    pl.close()
