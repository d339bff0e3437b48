"""TEST INFRASTRUCTURE (a Python model of the device-resident exchange; the product's implementation is native:
csrc/guber_global_sync.h).

GLOBAL behaviour across the GPUs of one node with every row staying in HBM (reference: global.go,
gubernator.go:395-459,510-512; host-staged twin: pyglobal.py, which also documents the semantics).

Per sync and rank:
  guber_global_take_dev(role hits)   pending hit rows, device arrays                       (= sendHits, global.go:144-187)
  guber_ring_route_rows_dev          owning GPU of every row (replicated_hash.go:104-119)
  stable partition by owner          torch.argsort — plumbing
  all_to_all_single                  rows to their owners over RCCL / xGMI (one exchange, variable splits)
  guber_eval_batch_dev               owner applies: IsOwner, GLOBAL => DRAIN_OVER_LIMIT     (gubernator.go:497-512)
  guber_global_take_dev(role update) + guber_eval_batch_dev(hits = 0)                      (= broadcastPeers, global.go:234-283)
  all_gather_into_tensor             item rows (UpdatePeerGlobals item construction, gubernator.go:425-459)
  guber_add_items_dev                every other rank installs them                        (gubernator.go:452)

torch is used for device memory, stream ordering, the byte-level (un)packing of rows and torch.distributed — the
engine's own kernels do the table work.  The engine must have been created on the torch stream that is current
when sync() runs (Engine(stream=torch.cuda.current_stream().cuda_stream)), so torch ops and engine kernels are
ordered by the stream.
"""
import ctypes as C

import torch

from gubernator_amd import GuberError, abi, lib
from gubernator_amd.abi import GuberBatch, GuberResult

GLOBAL, DRAIN_OVER_LIMIT = abi.GLOBAL, abi.DRAIN_OVER_LIMIT
ROLE_HITS, ROLE_UPDATE = 1, 2
ITEM_E_RETRY = 5


class GuberGlobalRowsDev(C.Structure):
    _fields_ = [("cap", C.c_uint32), ("key_stride", C.c_uint32), ("key_bytes", C.c_void_p), ("key_len", C.c_void_p),
                ("hits", C.c_void_p), ("limit", C.c_void_p), ("duration", C.c_void_p), ("burst", C.c_void_p),
                ("created_at", C.c_void_p), ("behavior", C.c_void_p), ("algorithm", C.c_void_p), ("role", C.c_void_p)]


class GuberItemsDev(C.Structure):
    _fields_ = [("n", C.c_uint32), ("reserved", C.c_uint32), ("key_bytes", C.c_void_p), ("key_off", C.c_void_p),
                ("algorithm", C.c_void_p), ("status", C.c_void_p), ("limit", C.c_void_p), ("duration", C.c_void_p),
                ("remaining", C.c_void_p), ("remaining_f", C.c_void_p), ("stamp", C.c_void_p), ("burst", C.c_void_p),
                ("expire_at", C.c_void_p), ("invalid_at", C.c_void_p)]


_bound = False


def _lib():
    global _bound
    L = lib()
    if not _bound:
        L.guber_global_pending.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.guber_global_take_dev.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(GuberGlobalRowsDev), C.POINTER(C.c_uint32)]
        L.guber_ring_route_rows_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.guber_add_items_dev.argtypes = [C.c_void_p, C.POINTER(GuberItemsDev), C.c_void_p]
        _bound = True
    return L


def _check(rc):
    if rc != 0:
        L = lib()
        raise GuberError(rc, f"{L.guber_strerror(rc).decode()} ({L.guber_last_error().decode()})")


def _bytes_of(t, width):
    """[n] tensor of `width`-byte elements -> [n, width] uint8 view"""
    return t.contiguous().view(torch.uint8).reshape(-1, width)


class DevRows:
    """Pending GLOBAL rows in HBM.  key i = key_mat[i, :key_len[i]].  u32 columns are held as int32 (same bits)."""
    I64 = ("hits", "limit", "duration", "burst", "created_at")

    def __init__(self, key_mat, key_len, behavior, hits, limit, duration, burst, created_at, algorithm, role):
        self.key_mat, self.key_len, self.behavior = key_mat, key_len, behavior
        self.hits, self.limit, self.duration, self.burst, self.created_at = hits, limit, duration, burst, created_at
        self.algorithm, self.role = algorithm, role

    def __len__(self):
        return int(self.key_len.shape[0])

    @staticmethod
    def alloc(n, stride, dev):
        z64 = lambda: torch.zeros(n, dtype=torch.int64, device=dev)
        return DevRows(torch.zeros((n, stride), dtype=torch.uint8, device=dev), torch.zeros(n, dtype=torch.int32, device=dev),
                       torch.zeros(n, dtype=torch.int32, device=dev), z64(), z64(), z64(), z64(), z64(),
                       torch.zeros(n, dtype=torch.uint8, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev))

    def head(self, n):
        return DevRows(self.key_mat[:n], self.key_len[:n], self.behavior[:n], self.hits[:n], self.limit[:n], self.duration[:n],
                       self.burst[:n], self.created_at[:n], self.algorithm[:n], self.role[:n])

    def select(self, idx):
        return DevRows(self.key_mat[idx], self.key_len[idx], self.behavior[idx], self.hits[idx], self.limit[idx], self.duration[idx],
                       self.burst[idx], self.created_at[idx], self.algorithm[idx], self.role[idx])

    # byte image of a row: key | key_len u32 | behavior u32 | 5 x i64 | algorithm u8 | role u8 | 6 pad
    def pack(self):
        n = len(self)
        pad = torch.zeros((n, 6), dtype=torch.uint8, device=self.key_mat.device)
        cols = [self.key_mat, _bytes_of(self.key_len, 4), _bytes_of(self.behavior, 4)]
        cols += [_bytes_of(getattr(self, c), 8) for c in DevRows.I64]
        cols += [self.algorithm.reshape(n, 1), self.role.reshape(n, 1), pad]
        return torch.cat(cols, dim=1).contiguous()

    @staticmethod
    def row_bytes(stride):
        return stride + 8 + 40 + 8

    @staticmethod
    def unpack(mat, stride):
        n = mat.shape[0]
        o = stride
        i32 = lambda a: mat[:, a:a + 4].contiguous().view(torch.int32).reshape(n)
        i64 = lambda a: mat[:, a:a + 8].contiguous().view(torch.int64).reshape(n)
        return DevRows(mat[:, :stride].contiguous(), i32(o), i32(o + 4), i64(o + 8), i64(o + 16), i64(o + 24), i64(o + 32), i64(o + 40),
                       mat[:, o + 48].contiguous(), mat[:, o + 49].contiguous())

    def packed_keys(self):
        """(key_bytes with 8 readable bytes past the end, key_off int32[n+1]) as guber_batch_t wants them"""
        n, stride = self.key_mat.shape
        mask = torch.arange(stride, device=self.key_mat.device, dtype=torch.int32)[None, :] < self.key_len[:, None]
        kb = torch.cat([self.key_mat[mask], torch.zeros(8, dtype=torch.uint8, device=self.key_mat.device)])
        ko = torch.zeros(n + 1, dtype=torch.int32, device=self.key_mat.device)
        ko[1:] = torch.cumsum(self.key_len, 0)
        return kb, ko

    def keys(self):
        km, kl = self.key_mat.cpu().numpy(), self.key_len.cpu().numpy()
        return [km[i, :int(kl[i])].tobytes() for i in range(len(kl))]


class DevResult:
    def __init__(self, n, dev):
        m = max(n, 1)
        self.status = torch.zeros(m, dtype=torch.uint8, device=dev)
        self.err = torch.zeros(m, dtype=torch.uint8, device=dev)
        self.limit, self.remaining, self.reset_time = (torch.zeros(m, dtype=torch.int64, device=dev) for _ in range(3))
        self.c = GuberResult(self.status.data_ptr(), self.limit.data_ptr(), self.remaining.data_ptr(), self.reset_time.data_ptr(),
                             self.err.data_ptr(), 0, 0, 0, 0, 0)


class GlobalSyncDev:
    """Per-rank half of the exchange.  node = gubernator_amd.Engine created with FLAG_GLOBAL, max_key_bytes = the
    node-wide key stride, on the current torch stream."""

    def __init__(self, node, rank, world, ring, transport, device, key_stride=64):
        self.node, self.rank, self.world, self.ring, self.transport = node, rank, world, ring, transport
        self.dev, self.stride = torch.device(device), (key_stride + 7) & ~7
        self.L = _lib()
        self.bytes_moved = 0
        self.fallbacks = 0

    def evaluate(self, keys, hits, limit, duration, now_ms, **kw):
        """A batch of GLOBAL requests arriving at this rank from clients (V1Instance.GetRateLimits,
        gubernator.go:247-270): owned keys are evaluated as the owner, the others against the local replica."""
        import numpy as np
        from gubernator_amd.abi import HostBatch
        hb = HostBatch(keys, hits, limit, duration, now_ms, **kw)
        owner = self.ring.route((hb.key_bytes, hb.key_off)) if hb.n else np.zeros(0, np.uint32)
        kw = dict(kw)
        kw["behavior"] = np.broadcast_to(np.asarray(kw.get("behavior", 0), np.uint32), (hb.n,)) | np.uint32(GLOBAL)
        return self.node.eval(HostBatch((hb.key_bytes, hb.key_off), hits, limit, duration, now_ms,
                                        is_owner=(owner == self.rank).astype(np.uint8), **kw))

    # ---- engine calls on device arrays ----------------------------------------------------------------
    def take(self, role_mask):
        n = C.c_uint32()
        _check(self.L.guber_global_pending(self.node.h, C.byref(n)))
        if n.value == 0:
            return DevRows.alloc(0, self.stride, self.dev)
        r = DevRows.alloc(n.value, self.stride, self.dev)
        torch.cuda.current_stream(self.dev).synchronize()          # zero-fill done before the engine writes (same stream: cheap)
        out = GuberGlobalRowsDev(n.value, self.stride, r.key_mat.data_ptr(), r.key_len.data_ptr(), r.hits.data_ptr(), r.limit.data_ptr(),
                                 r.duration.data_ptr(), r.burst.data_ptr(), r.created_at.data_ptr(), r.behavior.data_ptr(),
                                 r.algorithm.data_ptr(), r.role.data_ptr())
        m = C.c_uint32()
        _check(self.L.guber_global_take_dev(self.node.h, role_mask, C.byref(out), C.byref(m)))
        return r.head(m.value)

    def owners(self, rows):
        n = len(rows)
        owner = torch.zeros(n, dtype=torch.int32, device=self.dev)
        if n:
            _check(self.L.guber_ring_route_rows_dev(self.node.h, self.ring.h, rows.key_mat.data_ptr(), self.stride, rows.key_len.data_ptr(),
                                                    n, owner.data_ptr()))
        return owner

    def eval_rows(self, rows, now_ms, is_owner, zero_hits=False, drain=False):
        """One request per row (the aggregated RateLimitReq of global.go:100-111) through guber_eval_batch_dev."""
        n = len(rows)
        res = DevResult(n, self.dev)
        chunk = self.node.max_batch
        for lo in range(0, n, chunk):
            sub = rows.select(slice(lo, min(n, lo + chunk)))
            m = len(sub)
            kb, ko = sub.packed_keys()
            beh = sub.behavior
            if drain:   # GetPeerRateLimits: GLOBAL => DRAIN_OVER_LIMIT (gubernator.go:510-512)
                beh = torch.where((beh & GLOBAL) != 0, beh | DRAIN_OVER_LIMIT, beh)
            hits = torch.zeros_like(sub.hits) if zero_hits else sub.hits.contiguous()
            owner = torch.full((m,), 1 if is_owner else 0, dtype=torch.uint8, device=self.dev)
            cols = [sub.limit.contiguous(), sub.duration.contiguous(), sub.burst.contiguous(), sub.created_at.contiguous(),
                    sub.algorithm.contiguous(), beh.contiguous()]
            part = DevResult(m, self.dev)
            b = GuberBatch(m, 0, kb.data_ptr(), ko.data_ptr(), hits.data_ptr(), cols[0].data_ptr(), cols[1].data_ptr(), cols[2].data_ptr(),
                           cols[3].data_ptr(), cols[4].data_ptr(), cols[5].data_ptr(), owner.data_ptr(), None, None, now_ms)
            self.node.eval_dev(b, part.c)
            retry = part.err[:m] == ITEM_E_RETRY
            if bool(retry.any()):       # true 64-bit hash collision inside the batch: the host path resolves it (astronomically rare)
                self._eval_on_host(sub, retry, now_ms, is_owner, zero_hits, drain, part)
            for f in ("status", "err", "limit", "remaining", "reset_time"):
                getattr(res, f)[lo:lo + m] = getattr(part, f)[:m]
            self._keep = (kb, ko, hits, cols, owner, part)          # alive until the stream has consumed them
        return res

    def _eval_on_host(self, sub, retry, now_ms, is_owner, zero_hits, drain, part):
        import numpy as np
        from gubernator_amd.abi import HostBatch
        self.fallbacks += 1
        idx = torch.nonzero(retry).reshape(-1)
        r = sub.select(idx)
        keys = r.keys()
        beh = r.behavior.cpu().numpy().astype(np.uint32)
        if drain:
            beh = np.where(beh & GLOBAL, beh | DRAIN_OVER_LIMIT, beh).astype(np.uint32)
        hits = np.zeros(len(keys), np.int64) if zero_hits else r.hits.cpu().numpy()
        hb = HostBatch(keys, hits, r.limit.cpu().numpy(), r.duration.cpu().numpy(), now_ms, burst=r.burst.cpu().numpy(),
                       created_at=r.created_at.cpu().numpy(), algorithm=r.algorithm.cpu().numpy(), behavior=beh,
                       is_owner=np.full(len(keys), 1 if is_owner else 0, np.uint8))
        out = self.node.eval(hb)
        for f in ("status", "err", "limit", "remaining", "reset_time"):
            getattr(part, f)[idx] = torch.from_numpy(getattr(out, f)[:len(keys)].copy()).to(self.dev)

    # ---- item rows: key | key_len u32 | pad u32 | limit duration remaining i64 | remaining_f f64 | stamp burst expire_at i64 | algo status + 6 pad
    def item_row_bytes(self):
        return self.stride + 8 + 56 + 8

    def updates_to_item_rows(self, rows, res, now_ms):
        """UpdatePeerGlobals item construction (gubernator.go:425-459) from the owner's hits = 0 status, as packed rows."""
        n = len(rows)
        ok = res.err[:n] == 0                 # broadcastPeers skips keys whose status read failed (global.go:246-249)
        rows = rows.select(ok)
        status, limit, remaining, reset = (a[:n][ok] for a in (res.status, res.limit, res.remaining, res.reset_time))
        m = len(rows)
        leaky = rows.algorithm == abi.LEAKY
        zero64 = torch.zeros(m, dtype=torch.int64, device=self.dev)
        cols = [rows.key_mat, _bytes_of(rows.key_len, 4), torch.zeros((m, 4), dtype=torch.uint8, device=self.dev),
                _bytes_of(limit, 8), _bytes_of(rows.duration, 8), _bytes_of(torch.where(leaky, zero64, remaining), 8),
                _bytes_of(torch.where(leaky, remaining.to(torch.float64), torch.zeros(m, dtype=torch.float64, device=self.dev)), 8),
                _bytes_of(torch.full((m,), now_ms, dtype=torch.int64, device=self.dev), 8),
                _bytes_of(torch.where(leaky, limit, zero64), 8), _bytes_of(reset, 8),
                rows.algorithm.reshape(m, 1), torch.where(leaky, torch.zeros_like(status), status).reshape(m, 1),
                torch.zeros((m, 6), dtype=torch.uint8, device=self.dev)]
        return torch.cat(cols, dim=1).contiguous()

    def install_item_rows(self, mat):
        """AddCacheItem for every received global (UpdatePeerGlobals receiver side)."""
        n = mat.shape[0]
        if n == 0:
            return
        o = self.stride
        key_len = mat[:, o:o + 4].contiguous().view(torch.int32).reshape(n)
        i64 = lambda a: mat[:, a:a + 8].contiguous().view(torch.int64).reshape(n)
        limit, duration, remaining = i64(o + 8), i64(o + 16), i64(o + 24)
        remaining_f = mat[:, o + 32:o + 40].contiguous().view(torch.float64).reshape(n)
        stamp, burst, expire_at = i64(o + 40), i64(o + 48), i64(o + 56)
        algorithm, status = mat[:, o + 64].contiguous(), mat[:, o + 65].contiguous()
        key_mat = mat[:, :o]
        mask = torch.arange(o, device=self.dev, dtype=torch.int32)[None, :] < key_len[:, None]
        kb = torch.cat([key_mat[mask], torch.zeros(8, dtype=torch.uint8, device=self.dev)])
        ko = torch.zeros(n + 1, dtype=torch.int32, device=self.dev)
        ko[1:] = torch.cumsum(key_len, 0)
        result = torch.zeros(n, dtype=torch.uint8, device=self.dev)
        it = GuberItemsDev(n, 0, kb.data_ptr(), ko.data_ptr(), algorithm.data_ptr(), status.data_ptr(), limit.data_ptr(), duration.data_ptr(),
                           remaining.data_ptr(), remaining_f.data_ptr(), stamp.data_ptr(), burst.data_ptr(), expire_at.data_ptr(), None)
        _check(self.L.guber_add_items_dev(self.node.h, C.byref(it), result.data_ptr()))
        bad = result >= 0xFE
        if bool(bad.any()):               # in-call hash collision: hand those few to the host entry point
            self.fallbacks += 1
            import numpy as np
            from gubernator_amd import make_item
            idx = torch.nonzero(bad).reshape(-1).cpu().numpy()
            km, kl = key_mat.cpu().numpy(), key_len.cpu().numpy()
            h = {f: t.cpu().numpy() for f, t in dict(limit=limit, duration=duration, remaining=remaining, remaining_f=remaining_f,
                                                      stamp=stamp, burst=burst, expire_at=expire_at, algorithm=algorithm, status=status).items()}
            for i in idx.tolist():
                self.node.add_item(make_item(km[i, :int(kl[i])].tobytes(), int(h["algorithm"][i]), limit=int(h["limit"][i]),
                                             duration=int(h["duration"][i]), remaining=int(h["remaining"][i]),
                                             remaining_f=float(h["remaining_f"][i]), stamp=int(h["stamp"][i]), burst=int(h["burst"][i]),
                                             expire_at=int(h["expire_at"][i]), status=int(h["status"][i])), 0)
        self._keep2 = (kb, ko, result, limit, duration, remaining, remaining_f, stamp, burst, expire_at, algorithm, status)

    # ---- one GlobalSyncWait tick ----------------------------------------------------------------------
    def sync(self, now_ms):
        hits_rows = self.take(1 << ROLE_HITS)
        owner = self.owners(hits_rows)
        order = torch.argsort(owner, stable=True)
        counts = torch.bincount(owner, minlength=self.world)[:self.world].cpu().tolist() if len(hits_rows) else [0] * self.world
        send = hits_rows.pack()[order] if len(hits_rows) else torch.zeros((0, DevRows.row_bytes(self.stride)), dtype=torch.uint8, device=self.dev)
        recv = self.transport.exchange_rows(send, counts)             # rows for me, in source-rank order
        mine = DevRows.unpack(recv, self.stride)
        self.bytes_moved += int(recv.shape[0]) * int(recv.shape[1])
        if len(mine):
            self.eval_rows(mine, now_ms, True, drain=True)
        upd = self.take(1 << ROLE_UPDATE)
        if len(upd):
            items = self.updates_to_item_rows(upd, self.eval_rows(upd, now_ms, False, zero_hits=True), now_ms)
        else:
            items = torch.zeros((0, self.item_row_bytes()), dtype=torch.uint8, device=self.dev)
        everyone = self.transport.gather_rows(items)
        for src in range(self.world):
            if src != self.rank:
                self.bytes_moved += int(everyone[src].shape[0]) * int(everyone[src].shape[1])
                self.install_item_rows(everyone[src])
        return dict(hits_sent=len(hits_rows), hits_applied=len(mine), broadcast=int(items.shape[0]))


class TorchTransportDev:
    """Row exchange over torch.distributed on device tensors: backend nccl = RCCL over xGMI."""

    def __init__(self, device, group=None):
        import torch.distributed as dist
        self.dist, self.group, self.dev = dist, group, torch.device(device)
        self.world = dist.get_world_size(group)

    def _all_counts(self, mine):
        t = torch.tensor(mine, dtype=torch.int64, device=self.dev)
        out = torch.zeros(self.world * len(mine), dtype=torch.int64, device=self.dev)
        self.dist.all_gather_into_tensor(out, t, group=self.group)
        return out.reshape(self.world, len(mine)).cpu().tolist()

    def exchange_rows(self, send, counts):
        rank = self.dist.get_rank(self.group)
        table = self._all_counts(counts)                               # table[src][dst]
        out_splits = [table[src][rank] for src in range(self.world)]
        recv = torch.zeros((sum(out_splits), send.shape[1]), dtype=torch.uint8, device=self.dev)
        self.dist.all_to_all_single(recv, send.contiguous(), out_splits, list(counts), group=self.group)
        return recv

    def gather_rows(self, rows):
        ns = [c[0] for c in self._all_counts([int(rows.shape[0])])]
        mx = max(ns)
        pad = torch.zeros((mx, rows.shape[1]), dtype=torch.uint8, device=self.dev)
        pad[:rows.shape[0]] = rows
        out = torch.zeros((self.world * mx, rows.shape[1]), dtype=torch.uint8, device=self.dev)
        if mx:
            self.dist.all_gather_into_tensor(out, pad, group=self.group)
        return [out[r * mx:r * mx + ns[r]] for r in range(self.world)]


class LocalClusterDev:
    """N logical ranks in ONE process on one device (N engines on one torch stream): the same GlobalSyncDev code with
    the collectives replaced by in-memory hand-over, ranks stepped in lock-step by the caller's thread pool."""

    class _Transport:
        def __init__(self, cluster, rank):
            self.c, self.rank = cluster, rank

        def exchange_rows(self, send, counts):
            table = self.c._gather(self.rank, (send, counts))
            parts = []
            for src in range(self.c.world):
                s, cnt = table[src]
                lo = sum(cnt[:self.rank])
                parts.append(s[lo:lo + cnt[self.rank]])
            return torch.cat(parts) if parts else send[:0]

        def gather_rows(self, rows):
            return self.c._gather(self.rank, rows)

    def __init__(self, nodes, ring, device, key_stride=64):
        import threading
        self.world = len(nodes)
        self.ranks = [GlobalSyncDev(n, r, self.world, ring, LocalClusterDev._Transport(self, r), device, key_stride) for r, n in enumerate(nodes)]
        self._barrier = threading.Barrier(self.world)
        self._slots = [None] * self.world
        self._threading = threading
        self.dev = torch.device(device)

    def _gather(self, rank, obj):
        torch.cuda.current_stream(self.dev).synchronize()
        self._slots[rank] = obj
        self._barrier.wait()
        out = list(self._slots)
        self._barrier.wait()
        return out

    def sync(self, now_ms):
        out, errs = [None] * self.world, []
        stream = torch.cuda.current_stream(self.dev)

        def run(r):
            try:
                with torch.cuda.stream(stream):
                    out[r] = self.ranks[r].sync(now_ms)
            except BaseException as ex:   # noqa: BLE001
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
