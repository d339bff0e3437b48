"""GLOBAL behaviour across the GPUs of one node (reference: global.go, gubernator.go:395-459,510-512).

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

from . import abi
from .abi import HostBatch

GLOBAL, RESET_REMAINING, DRAIN_OVER_LIMIT = abi.GLOBAL, abi.RESET_REMAINING, abi.DRAIN_OVER_LIMIT
ROLE_HITS, ROLE_UPDATE = 1, 2

# numpy image of guber_item_t (include/guber_gpu.h), 80 bytes
ITEM_DTYPE = np.dtype({"names": ["algorithm", "status", "reserved0", "key_len", "key", "limit", "duration", "remaining",
                                 "remaining_f", "stamp", "burst", "expire_at", "invalid_at"],
                       "formats": ["u1", "u1", "u2", "u4", "u8", "i8", "i8", "i8", "f8", "i8", "i8", "i8", "i8"],
                       "offsets": [0, 1, 2, 4, 8, 16, 24, 32, 40, 48, 56, 64, 72], "itemsize": 80})
assert ITEM_DTYPE.itemsize == C.sizeof(abi.GuberItem)


class Rows:
    """Pending GLOBAL rows, structure of arrays.  key i = key_mat[i, :key_len[i]]."""
    COLS = ("key_len", "hits", "limit", "duration", "burst", "created_at", "behavior", "algorithm", "role")

    def __init__(self, key_mat, key_len, hits, limit, duration, burst, created_at, behavior, algorithm, role):
        self.key_mat = key_mat
        self.key_len, self.hits, self.limit, self.duration, self.burst = key_len, hits, limit, duration, burst
        self.created_at, self.behavior, self.algorithm, self.role = created_at, behavior, algorithm, role

    @staticmethod
    def empty(stride=8):
        z = lambda dt: np.zeros(0, dt)
        return Rows(np.zeros((0, stride), np.uint8), z(np.uint32), z(np.int64), z(np.int64), z(np.int64), z(np.int64),
                    z(np.int64), z(np.uint32), z(np.uint8), z(np.uint8))

    @staticmethod
    def from_dicts(rows, stride=64):
        n = len(rows)
        stride = max([stride] + [len(r["key"]) for r in rows])
        km = np.zeros((n, stride), np.uint8)
        for i, r in enumerate(rows):
            km[i, :len(r["key"])] = np.frombuffer(r["key"], np.uint8)
        col = lambda f, dt: np.array([r[f] for r in rows], dtype=dt)
        return Rows(km, np.array([len(r["key"]) for r in rows], np.uint32), col("hits", np.int64), col("limit", np.int64),
                    col("duration", np.int64), col("burst", np.int64), col("created_at", np.int64),
                    col("behavior", np.uint32), col("algorithm", np.uint8), col("role", np.uint8))

    def __len__(self):
        return len(self.key_len)

    def select(self, idx):
        return Rows(self.key_mat[idx], *[getattr(self, c)[idx] for c in Rows.COLS])

    @staticmethod
    def concat(parts):
        parts = [p for p in parts if len(p)]
        if not parts:
            return Rows.empty()
        stride = max(p.key_mat.shape[1] for p in parts)
        mats = [np.pad(p.key_mat, ((0, 0), (0, stride - p.key_mat.shape[1]))) for p in parts]
        return Rows(np.concatenate(mats), *[np.concatenate([getattr(p, c) for p in parts]) for c in Rows.COLS])

    def packed_keys(self):
        """(key_bytes, key_off) as the C ABI wants them (8 readable bytes past the end)."""
        n, stride = self.key_mat.shape
        mask = np.arange(stride, dtype=np.uint32)[None, :] < self.key_len[:, None]
        kb = np.concatenate([self.key_mat[mask], np.zeros(8, np.uint8)])
        ko = np.zeros(n + 1, np.uint32)
        np.cumsum(self.key_len, out=ko[1:])
        return kb, ko

    def keys(self):
        return [self.key_mat[i, :int(self.key_len[i])].tobytes() for i in range(len(self))]

    def nbytes(self):
        return int(self.key_len.sum()) + 53 * len(self)


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
