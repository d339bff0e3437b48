"""Sharding one node's GPUs the way gubernator shards peers: every GPU is a "peer" on the reference's
replicated consistent hash (replicated_hash.go:29-119) and owns the keys the ring assigns to it.  The
data path needs no collective — each rank evaluates the requests for the keys it owns;
torch.distributed is used only for the barrier / max-over-ranks timing and, in tests, to check that
the ranks' partitions are disjoint and complete.
"""
import numpy as np

from . import Ring


def peer_names(world):
    return [f"gpu{i}" for i in range(world)]


def owned_key_ids(table, world, rank, route=None, chunk=2_000_000, replicas=512, hash_kind="fnv1"):
    """ids (rows of the fixed-width key `table`) owned by `rank` among `world` peers.
    route(key_bytes, key_off) -> owner array; default = host ring lookup (guber_ring_route).  bench.py
    passes the device router (k_route) instead."""
    total = table.shape[0]
    if world == 1:
        return np.arange(total, dtype=np.int64)
    ring = Ring(peer_names(world), replicas, hash_kind)
    L = table.shape[1]
    owned = []
    for lo in range(0, total, chunk):
        hi = min(lo + chunk, total)
        kb = np.concatenate([np.ascontiguousarray(table[lo:hi]).reshape(-1), np.zeros(8, np.uint8)])
        ko = (np.arange(hi - lo + 1, dtype=np.uint64) * L).astype(np.uint32)
        owner = route(ring, kb, ko) if route is not None else ring.route((kb, ko))
        owned.append(np.nonzero(owner == rank)[0].astype(np.int64) + lo)
    ring.close()
    return np.concatenate(owned) if owned else np.zeros(0, np.int64)


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (timing contract of bench.py); identity when not distributed."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


class SlotMap:
    """Load-aware placement of one GPU's keys on its logical shards.

    The logical shards inside a GPU are this engine's construct (the reference's workers are goroutines over hash ranges,
    workers.go:125-184; which worker holds a key never shows in a response), so the front end is free to place keys where
    the load is even.  It matters: every shard is a serial chain of batches, and with a skewed stream the shard that owns
    the hottest key becomes the chain everybody waits for (Zipf-1.1 over 10 M keys: one key is 11.6 % of the requests;
    on a plain consistent hash over 12 shards its shard carries 19 % of the stream, the others 7.4 % each).

    Keys map to `n_slots` hash slots (the reference's replicated consistent hash over slot names: guber_ring_*, k_route on
    the device), slots map to shards through a table, and keys that alone weigh more than a fraction of a shard's fair
    share are placed individually (hot-key isolation).  `place()` fills table and exception list from observed traffic
    with longest-processing-time-first: heaviest item to the least loaded shard.  Nothing here changes any result."""

    def __init__(self, n_shards, n_slots=256, vnodes=32, hash_kind="fnv1", prefix="slot"):
        if n_slots * vnodes > 8192:
            raise ValueError("the slot ring must fit the device router's LDS (n_slots * vnodes <= 8192)")
        self.n_shards, self.n_slots = int(n_shards), int(n_slots)
        self.ring = Ring([f"{prefix}{i}" for i in range(n_slots)], vnodes, hash_kind)
        self.table = (np.arange(n_slots) % n_shards).astype(np.uint8 if n_shards <= 256 else np.uint16)   # before any traffic: round robin
        self.hot_ids = np.zeros(0, np.int64)            # key ids placed individually ...
        self.hot_shard = np.zeros(0, self.table.dtype)  # ... and where
        self.load = np.zeros(n_shards)

    def place(self, slot_of_key, observed_ids, heavy_fraction=0.125):
        """slot_of_key[i] = slot of key id i; observed_ids = key ids of a sample of the request stream.
        -> shard of every key id."""
        n_keys = len(slot_of_key)
        counts = np.bincount(observed_ids, minlength=n_keys).astype(np.float64)
        total = max(counts.sum(), 1.0)
        fair = total / self.n_shards
        heavy = np.nonzero(counts > fair * heavy_fraction)[0]
        light = counts.copy()
        light[heavy] = 0.0
        # a slot weighs what its keys were seen to carry, plus a little per resident key so that silent slots spread too
        slot_w = np.bincount(slot_of_key, weights=light, minlength=self.n_slots) + \
            np.bincount(slot_of_key, minlength=self.n_slots) * (0.05 * total / max(n_keys, 1))
        items = [(float(counts[k]), 0, int(k)) for k in heavy] + [(float(slot_w[s]), 1, int(s)) for s in range(self.n_slots)]
        items.sort(key=lambda t: (-t[0], t[1], t[2]))
        load = np.zeros(self.n_shards)
        hot_ids, hot_shard = [], []
        for w, kind, x in items:
            j = int(np.argmin(load))
            load[j] += w
            if kind == 0:
                hot_ids.append(x); hot_shard.append(j)
            else:
                self.table[x] = j
        self.hot_ids = np.asarray(hot_ids, np.int64)
        self.hot_shard = np.asarray(hot_shard, self.table.dtype)
        self.load = load / total
        return self.shard_of(slot_of_key)

    def shard_of(self, slot_of_key, key_ids=None):
        """shard of keys given their slots (and, for the exception list, their ids: default = ids 0..n-1)"""
        out = self.table[slot_of_key]
        if len(self.hot_ids):
            if key_ids is None:
                out[self.hot_ids] = self.hot_shard
            else:
                pos = {int(k): int(s) for k, s in zip(self.hot_ids, self.hot_shard)}
                for q in np.nonzero(np.isin(key_ids, self.hot_ids))[0]:
                    out[q] = pos[int(key_ids[q])]
        return out

    def close(self):
        self.ring.close()
