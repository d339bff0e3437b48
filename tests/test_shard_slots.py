"""shard.SlotMap: load-aware placement of a GPU's keys on its logical shards (hash slots + hot-key isolation)."""
import numpy as np

import gubernator_amd as ga
import streams
from gubernator_amd import shard


def _slots(sm, nk):
    tab = streams.key_table(nk)
    kb, ko = streams.keys_for_ids(tab, np.arange(nk))
    return sm.ring.route((kb, ko)).astype(np.int64), (kb, ko)


def test_slotmap_balances_a_zipf_stream_and_isolates_the_hot_key():
    nk, S = 200_000, 12
    sm = shard.SlotMap(S)
    slot_of, keys = _slots(sm, nk)
    assert slot_of.min() >= 0 and slot_of.max() < sm.n_slots
    observed = streams.ZipfSampler(nk, s=1.1, seed=7).draw(1 << 20)
    sown = sm.place(slot_of, observed)
    assert sown.shape == (nk,) and sown.min() >= 0 and sown.max() < S
    later = streams.ZipfSampler(nk, s=1.1, seed=1234).draw(1 << 21)           # the stream that is measured later
    share = np.bincount(sown[later], minlength=S) / len(later)
    hot = np.bincount(later).max() / len(later)
    assert share.max() <= max(hot, 1.0 / S) * 1.08, share                     # no shard above the hottest key's own share
    rest = np.sort(share)[:-1]
    assert rest.max() / rest.min() < 1.15, share                              # the others are even
    # a plain consistent hash over the shards leaves the hot key's shard with its 1/S of everything else on top
    ring = ga.Ring([f"s{j}" for j in range(S)], 512, "fnv1")
    plain = np.bincount(ring.route(keys)[later], minlength=S) / len(later)
    ring.close()
    assert plain.max() > share.max() * 1.2, (plain, share)
    # the hottest key sits (almost) alone; every key has exactly one shard; placement is deterministic
    top = int(np.bincount(later).argmax())
    assert top in set(sm.hot_ids.tolist())
    sm2 = shard.SlotMap(S)
    assert np.array_equal(sm2.place(slot_of, observed), sown)
    sm.close(); sm2.close()


def test_slotmap_without_traffic_spreads_keys_evenly():
    nk, S = 100_000, 8
    sm = shard.SlotMap(S)
    slot_of, _ = _slots(sm, nk)
    sown = sm.place(slot_of, np.zeros(0, np.int64))
    per = np.bincount(sown, minlength=S)
    assert per.min() > 0 and per.max() / per.min() < 1.2, per
    assert len(sm.hot_ids) == 0
    sub = np.array([5, 17, 99_999])
    assert np.array_equal(sm.shard_of(slot_of[sub], key_ids=sub), sown[sub])
    sm.close()
