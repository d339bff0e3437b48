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


def gather_over_ranks(value, device=None):
    """[value of rank 0, value of rank 1, ...] (python ints) on every rank; [value] when not distributed."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [int(value)]
    t = torch.tensor([int(value)], dtype=torch.int64, device=device if device is not None else "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [int(x.item()) for x in out]
