"""TEST INFRASTRUCTURE (a Python model of the exchange; the product's implementation is native: csrc/guber_global_sync.h).

GLOBAL behaviour across the GPUs of one node (reference: global.go, gubernator.go:395-459,510-512).

Every GPU ("peer") holds a replica of each GLOBAL bucket and answers requests from it immediately;
hits on non-owned keys are accumulated on the device (guber_global_take, role 1) and, at every sync,
shipped to the owning GPU, which applies them with DRAIN_OVER_LIMIT and then broadcasts the bucket's
state (role 2) to all other GPUs, where it replaces the replica (UpdatePeerGlobals).

GlobalSync is the per-rank half of that exchange, array-based end to end (no per-row Python).  It is
generic over
  node      : eval(HostBatch) -> HostResult, global_take(role_mask) -> Rows, add_items_struct(array)
              (gubernator_amd.Engine)
  transport : all_gather(obj) -> [obj per rank]   (TorchTransport = torch.distributed: RCCL over xGMI with
              backend "nccl", gloo on CPU; LocalCluster = several logical ranks inside one process)
"""
import ctypes as C

import numpy as np

from gubernator_amd import abi
from gubernator_amd.abi import HostBatch
from gubernator_amd.rows import ITEM_DTYPE, Rows  # noqa: F401

GLOBAL, RESET_REMAINING, DRAIN_OVER_LIMIT = abi.GLOBAL, abi.RESET_REMAINING, abi.DRAIN_OVER_LIMIT
ROLE_HITS, ROLE_UPDATE = 1, 2


def rows_to_batch(rows, now_ms, is_owner, zero_hits=False, drain=False):
    """One request per pending row (the aggregated RateLimitReq the reference sends, global.go:100-111)."""
    n = len(rows)
    beh = rows.behavior
    if drain:   # GetPeerRateLimits: GLOBAL => DRAIN_OVER_LIMIT (gubernator.go:510-512)
        beh = np.where(beh & GLOBAL, beh | DRAIN_OVER_LIMIT, beh).astype(np.uint32)
    return HostBatch(rows.packed_keys(), np.zeros(n, np.int64) if zero_hits else rows.hits, rows.limit, rows.duration, now_ms,
                     burst=rows.burst, created_at=rows.created_at, algorithm=rows.algorithm, behavior=beh,
                     is_owner=np.full(n, 1 if is_owner else 0, np.uint8))


def updates_to_items(rows, res, now_ms):
    """UpdatePeerGlobals item construction (gubernator.go:425-459) from the owner's hits=0 status:
    (structured item array without key pointers, packed key bytes, key offsets)."""
    n = len(rows)
    ok = res.err[:n] == 0                 # broadcastPeers skips keys whose status read failed (global.go:246-249)
    rows = rows.select(np.nonzero(ok)[0])
    status, limit, remaining, reset = (a[:n][ok] for a in (res.status, res.limit, res.remaining, res.reset_time))
    m = len(rows)
    it = np.zeros(m, ITEM_DTYPE)
    leaky = rows.algorithm == abi.LEAKY
    it["algorithm"] = rows.algorithm
    it["status"] = np.where(leaky, 0, status)
    it["key_len"] = rows.key_len
    it["limit"] = limit
    it["duration"] = rows.duration
    it["remaining"] = np.where(leaky, 0, remaining)
    it["remaining_f"] = np.where(leaky, remaining.astype(np.float64), 0.0)
    it["burst"] = np.where(leaky, limit, 0)
    it["stamp"] = now_ms
    it["expire_at"] = reset
    kb, ko = rows.packed_keys()
    return it, kb, ko


def install_items(node, it, kb, ko):
    """AddCacheItem for every received global (UpdatePeerGlobals receiver side)."""
    if len(it) == 0:
        return
    it = it.copy()
    kb = np.ascontiguousarray(kb)
    it["key"] = kb.ctypes.data + ko[:-1].astype(np.uint64)
    node.add_items_struct(it, keepalive=kb)


class GlobalSync:
    def __init__(self, node, rank, world, ring, transport):
        self.node, self.rank, self.world, self.ring, self.transport = node, rank, world, ring, transport
        self.bytes_moved = 0

    def owners(self, rows):
        if len(rows) == 0:
            return np.zeros(0, np.uint32)
        return self.ring.route(rows.packed_keys())

    def evaluate(self, keys, hits, limit, duration, now_ms, **kw):
        """A batch of GLOBAL requests arriving at this rank (V1Instance.GetRateLimits, gubernator.go:247-270):
        owned keys are evaluated as the owner, the others against the local replica."""
        hb = HostBatch(keys, hits, limit, duration, now_ms, **kw)
        owner = self.ring.route((hb.key_bytes, hb.key_off)) if hb.n else np.zeros(0, np.uint32)
        kw = dict(kw)
        kw["behavior"] = np.broadcast_to(np.asarray(kw.get("behavior", 0), np.uint32), (hb.n,)) | np.uint32(GLOBAL)
        return self.node.eval(HostBatch((hb.key_bytes, hb.key_off), hits, limit, duration, now_ms,
                                        is_owner=(owner == self.rank).astype(np.uint8), **kw))

    def sync(self, now_ms):
        """One GlobalSyncWait tick: flush hits to owners, owners apply and broadcast."""
        # --- sendHits (global.go:144-187): pending hits grouped by owning peer
        hits_rows = self.node.global_take(1 << ROLE_HITS)
        owner = self.owners(hits_rows)
        outbox = [hits_rows.select(np.nonzero(owner == dst)[0]) for dst in range(self.world)]
        inbox = self.transport.all_gather(outbox)                   # [source rank][dest rank] -> Rows
        mine = Rows.concat([inbox[src][self.rank] for src in range(self.world)])   # applied in source-rank order
        self.bytes_moved += mine.nbytes()
        chunk = getattr(self.node, "max_batch", 1 << 16)
        for lo in range(0, len(mine), chunk):
            # GetPeerRateLimits on the owner: IsOwner = true, GLOBAL => DRAIN_OVER_LIMIT (gubernator.go:497-512)
            self.node.eval(rows_to_batch(mine.select(slice(lo, lo + chunk)), now_ms, True, drain=True))
        # --- broadcastPeers (global.go:234-283): status with Hits = 0, then UpdatePeerGlobals everywhere else
        upd = self.node.global_take(1 << ROLE_UPDATE)
        items = (np.zeros(0, ITEM_DTYPE), np.zeros(8, np.uint8), np.zeros(1, np.uint32))
        if len(upd):
            parts = []
            for lo in range(0, len(upd), chunk):
                sub = upd.select(slice(lo, lo + chunk))
                parts.append(updates_to_items(sub, self.node.eval(rows_to_batch(sub, now_ms, False, zero_hits=True)), now_ms))
            kbs = [p[1][:-8] for p in parts]
            offs, base = [np.zeros(1, np.uint32)], 0
            for p in parts:
                offs.append(p[2][1:] + np.uint32(base)); base += int(p[2][-1])
            items = (np.concatenate([p[0] for p in parts]), np.concatenate(kbs + [np.zeros(8, np.uint8)]), np.concatenate(offs))
        everyone = self.transport.all_gather(items)
        for src in range(self.world):
            if src != self.rank:
                it, kb, ko = everyone[src]
                self.bytes_moved += int(len(kb)) + 49 * len(it)
                install_items(self.node, it, kb, ko)
        return dict(hits_sent=len(hits_rows), hits_applied=len(mine), broadcast=len(items[0]))


class TorchTransport:
    """all_gather of (pickled) numpy bundles over torch.distributed (backend nccl = RCCL over xGMI, or gloo)."""

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
        errs = []

        def run(r):
            try:
                out[r] = self.ranks[r].sync(now_ms)
            except BaseException as ex:   # noqa: BLE001 - surface failures of any rank to the caller
                errs.append(ex)
                self._barrier.abort()
        ts = [self._threading.Thread(target=run, args=(r,)) for r in range(self.world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise errs[0]
        return out
