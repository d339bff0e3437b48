"""GLOBAL behaviour across the GPUs of one node (reference: global.go, gubernator.go:395-459,510-512).

Every GPU ("peer") holds a replica of each GLOBAL bucket and answers requests from it immediately;
hits on non-owned keys are accumulated on the device (guber_global_take, role 1) and, at every sync,
shipped to the owning GPU, which applies them with DRAIN_OVER_LIMIT and then broadcasts the bucket's
state (role 2) to all other GPUs, where it replaces the replica (UpdatePeerGlobals).

GlobalSync is the per-rank half of that exchange.  It is generic over
  node      : eval(HostBatch) -> HostResult, global_take() -> rows, add_items(items)   (gubernator_amd.Engine)
  transport : all_gather(obj) -> [obj per rank]   (TorchTransport = torch.distributed: RCCL over xGMI with
              backend "nccl", gloo on CPU; LocalTransport = several logical ranks inside one process)
"""
import numpy as np

from . import abi
from .abi import HostBatch, make_item

GLOBAL, RESET_REMAINING, DRAIN_OVER_LIMIT = abi.GLOBAL, abi.RESET_REMAINING, abi.DRAIN_OVER_LIMIT


def owner_of(ring, keys):
    return ring.route(keys) if len(keys) else np.zeros(0, np.uint32)


def rows_to_batch(rows, now_ms, is_owner, hits_override=None, extra_behavior=0):
    """One request per pending row (the aggregated RateLimitReq the reference sends, global.go:100-111)."""
    n = len(rows)
    return HostBatch([r["key"] for r in rows],
                     [r["hits"] if hits_override is None else hits_override for r in rows],
                     [r["limit"] for r in rows], [r["duration"] for r in rows], now_ms,
                     burst=[r["burst"] for r in rows], created_at=[r["created_at"] for r in rows],
                     algorithm=[r["algorithm"] for r in rows],
                     behavior=[r["behavior"] | (extra_behavior if (r["behavior"] & GLOBAL) else 0) for r in rows],
                     is_owner=[1 if is_owner else 0] * n)


def updates_to_items(rows, res, now_ms):
    """UpdatePeerGlobals item construction (gubernator.go:425-459) from the owner's hits=0 status."""
    items = []
    for i, r in enumerate(rows):
        status, limit, remaining, reset_time, err = res.rows()[i]
        if err:
            continue   # broadcastPeers skips keys whose status read failed (global.go:246-249)
        if r["algorithm"] == abi.LEAKY:
            items.append(dict(key=r["key"], algorithm=abi.LEAKY, limit=limit, duration=r["duration"],
                              remaining_f=float(remaining), burst=limit, stamp=now_ms, expire_at=reset_time))
        else:
            items.append(dict(key=r["key"], algorithm=abi.TOKEN, status=status, limit=limit, duration=r["duration"],
                              remaining=remaining, stamp=now_ms, expire_at=reset_time))
    return items


def install_items(node, items):
    if items:
        node.add_items([make_item(it["key"], it["algorithm"], limit=it["limit"], duration=it["duration"],
                                  remaining=it.get("remaining", 0), remaining_f=it.get("remaining_f", 0.0),
                                  stamp=it["stamp"], burst=it.get("burst", 0), expire_at=it["expire_at"],
                                  status=it.get("status", 0)) for it in items])


class GlobalSync:
    def __init__(self, node, rank, world, ring, transport):
        self.node, self.rank, self.world, self.ring, self.transport = node, rank, world, ring, transport
        self.bytes_moved = 0

    def evaluate(self, keys, hits, limit, duration, now_ms, **kw):
        """A batch of GLOBAL requests arriving at this rank (V1Instance.GetRateLimits, gubernator.go:247-270):
        owned keys are evaluated as the owner, the others against the local replica."""
        owner = owner_of(self.ring, keys)
        kw = dict(kw)
        kw["behavior"] = np.broadcast_to(np.asarray(kw.get("behavior", 0), np.uint32), (len(keys),)) | np.uint32(GLOBAL)
        return self.node.eval(HostBatch(keys, hits, limit, duration, now_ms,
                                        is_owner=(owner == self.rank).astype(np.uint8), **kw))

    def sync(self, now_ms):
        """One GlobalSyncWait tick: flush hits to owners, owners apply and broadcast."""
        rows = self.node.global_take()
        hits_rows = [r for r in rows if r["role"] == 1]
        upd = {r["key"]: r for r in rows if r["role"] == 2}
        # --- sendHits (global.go:144-187): group by owning peer
        owner = owner_of(self.ring, [r["key"] for r in hits_rows])
        outbox = [[] for _ in range(self.world)]
        for r, o in zip(hits_rows, owner):
            outbox[int(o)].append(r)
        inbox = self.transport.all_gather(outbox)                   # [source rank][dest rank] -> rows
        mine = [r for src in range(self.world) for r in inbox[src][self.rank]]   # applied in source-rank order
        self.bytes_moved += sum(len(r["key"]) + 53 for r in mine)
        if mine:
            # GetPeerRateLimits on the owner: IsOwner = true, GLOBAL => DRAIN_OVER_LIMIT (gubernator.go:497-512)
            self.node.eval(rows_to_batch(mine, now_ms, True, extra_behavior=DRAIN_OVER_LIMIT))
        for r in self.node.global_take():
            if r["role"] == 2:
                upd[r["key"]] = r
        # --- broadcastPeers (global.go:234-283): status with Hits = 0, then UpdatePeerGlobals everywhere else
        urows = list(upd.values())
        items = []
        if urows:
            res = self.node.eval(rows_to_batch(urows, now_ms, False, hits_override=0))
            items = updates_to_items(urows, res, now_ms)
        everyone = self.transport.all_gather(items)
        for src in range(self.world):
            if src != self.rank:
                self.bytes_moved += sum(len(it["key"]) + 49 for it in everyone[src])
                install_items(self.node, everyone[src])
        return dict(hits_sent=len(hits_rows), hits_applied=len(mine), broadcast=len(items))


class TorchTransport:
    """all_gather of small python objects over torch.distributed (backend nccl = RCCL over xGMI, or gloo)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group

    def all_gather(self, obj):
        out = [None] * self.dist.get_world_size(self.group)
        self.dist.all_gather_object(out, obj, group=self.group)
        return out


class LocalCluster:
    """N logical ranks in ONE process (N engines on one or several devices): the same GlobalSync code, with
    the collectives replaced by in-memory hand-over.  Used where only one GPU is visible, and by the tests."""

    class _Transport:
        def __init__(self, cluster, rank):
            self.c, self.rank = cluster, rank

        def all_gather(self, obj):
            return self.c._gather(self.rank, obj)

    def __init__(self, nodes, ring):
        import threading
        self.world = len(nodes)
        self.ranks = [GlobalSync(n, r, self.world, ring, LocalCluster._Transport(self, r)) for r, n in enumerate(nodes)]
        self._barrier = threading.Barrier(self.world)
        self._slots = [None] * self.world
        self._threading = threading

    def _gather(self, rank, obj):
        self._slots[rank] = obj
        self._barrier.wait()
        out = list(self._slots)
        self._barrier.wait()
        return out

    def sync(self, now_ms):
        out = [None] * self.world

        def run(r):
            out[r] = self.ranks[r].sync(now_ms)
        ts = [self._threading.Thread(target=run, args=(r,)) for r in range(self.world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return out
