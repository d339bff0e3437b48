"""Reference model of GLOBAL behaviour for the tests: N peers, each a CPU oracle plus the host-side
queues of the reference's globalManager, written to follow global.go / gubernator.go line by line:
  request on a non-owner  getGlobalRateLimit  gubernator.go:395-421  (evaluate on the replica, QueueHit)
  request on the owner    getLocalRateLimit   gubernator.go:588-612  (evaluate, QueueUpdate)
  runAsyncHits / sendHits global.go:91-187    (sum Hits per key, OR RESET_REMAINING, first request = template)
  GetPeerRateLimits       gubernator.go:462-540 (IsOwner, GLOBAL => DRAIN_OVER_LIMIT)
  runBroadcasts / broadcastPeers global.go:193-283 (last request = template, status with Hits = 0)
  UpdatePeerGlobals       gubernator.go:425-459 (replica item construction)
Peers' flushes are applied in rank order (the reference's order is nondeterministic).
Also OracleNode: an oracle with the SAME pending-queue behaviour as the HIP engine (guber_global_take),
so the product orchestrator (tests/pyglobal.py) can be exercised on a machine without a GPU."""
import numpy as np

import support
from support import HostBatch

GLOBAL, RESET, DRAIN = 2, 8, 32


def mk_batch(reqs, now_ms, is_owner):
    return HostBatch([r["key"] for r in reqs], [r["hits"] for r in reqs], [r["limit"] for r in reqs],
                     [r["duration"] for r in reqs], now_ms, burst=[r.get("burst", 0) for r in reqs],
                     created_at=[r.get("created_at", now_ms) for r in reqs], algorithm=[r["algorithm"] for r in reqs],
                     behavior=[r["behavior"] for r in reqs], is_owner=[1 if is_owner else 0] * len(reqs))


class GlobalModel:
    def __init__(self, n, owner_fn):
        self.n, self.owner_fn = n, owner_fn
        self.oracles = [support.Oracle(cache_size=1 << 20) for _ in range(n)]
        self.hits = [dict() for _ in range(n)]      # global.go:93  hits := make(map[string]*RateLimitReq)
        self.updates = [dict() for _ in range(n)]   # global.go:195

    def request(self, rank, req, now_ms):
        req = dict(req)
        req.setdefault("created_at", now_ms)
        owner = self.owner_fn(req["key"])
        row = self.oracles[rank].eval(mk_batch([req], now_ms, owner == rank)).rows()[0]
        if row[4] == 0 and req["hits"] != 0:
            if owner == rank:
                self.updates[rank][req["key"]] = req                      # QueueUpdate: last wins (global.go:200)
            else:
                h = self.hits[rank]
                if req["key"] in h:                                       # global.go:100-111
                    if req["behavior"] & RESET:
                        h[req["key"]]["behavior"] |= RESET
                    h[req["key"]]["hits"] = support_wrap(h[req["key"]]["hits"] + req["hits"])
                else:
                    h[req["key"]] = req
        return row

    def sync(self, now_ms):
        for src in range(self.n):                                         # sendHits, in rank order
            for key, r in self.hits[src].items():
                o = self.owner_fn(key)
                r2 = dict(r)
                if r2["behavior"] & GLOBAL:
                    r2["behavior"] |= DRAIN                                # gubernator.go:510-512
                row = self.oracles[o].eval(mk_batch([r2], now_ms, True)).rows()[0]
                if row[4] == 0 and r2["hits"] != 0 and (r2["behavior"] & GLOBAL):
                    self.updates[o][key] = r2
            self.hits[src] = {}
        for o in range(self.n):                                           # broadcastPeers
            for key, u in self.updates[o].items():
                q = dict(u, hits=0)
                status, limit, remaining, reset_time, err = self.oracles[o].eval(mk_batch([q], now_ms, False)).rows()[0]
                if err:
                    continue
                if u["algorithm"] == 1:
                    it = support.make_item(key, 1, limit=limit, duration=u["duration"], remaining_f=float(remaining),
                                           burst=limit, stamp=now_ms, expire_at=reset_time)
                else:
                    it = support.make_item(key, 0, limit=limit, duration=u["duration"], remaining=remaining,
                                           stamp=now_ms, expire_at=reset_time, status=status)
                for p in range(self.n):
                    if p != o:
                        self.oracles[p].add_item(it, now_ms)
            self.updates[o] = {}


def support_wrap(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


class OracleNode:
    """eval / add_items / global_take with the engine's semantics (guber_kernels.h queue_global)."""

    def __init__(self):
        self.o = support.Oracle(cache_size=1 << 20)
        self.pending = {}

    def eval(self, batch):
        res = self.o.eval(batch)
        for i in range(batch.n):
            beh, hits = int(batch.behavior[i]), int(batch.hits[i])
            if not (beh & GLOBAL) or hits == 0 or res.err[i] != 0:
                continue
            off = batch.key_off
            key = batch.key_bytes[off[i]:off[i + 1]].tobytes()
            tmpl = dict(key=key, hits=hits, limit=int(batch.limit[i]), duration=int(batch.duration[i]),
                        burst=int(batch.burst[i]) if batch.burst is not None else 0,
                        created_at=int(batch.created_at[i]) if batch.created_at is not None else batch.now_ms,
                        behavior=beh, algorithm=int(batch.algorithm[i]))
            p = self.pending.get(key)
            owner = int(batch.is_owner[i]) if batch.is_owner is not None else 1
            if owner:
                self.pending[key] = dict(tmpl, hits=0, role=2)
            elif p is not None and p["role"] == 1:
                p["hits"] = support_wrap(p["hits"] + hits)
                p["behavior"] |= beh & RESET
            else:
                self.pending[key] = dict(tmpl, role=1)
        return res

    def add_items_struct(self, items, keepalive=None):
        import ctypes as C
        for i in range(len(items)):
            r = items[i]
            key = C.string_at(int(r["key"]), int(r["key_len"]))
            self.o.add_item(support.make_item(key, int(r["algorithm"]), limit=int(r["limit"]), duration=int(r["duration"]),
                                              remaining=int(r["remaining"]), remaining_f=float(r["remaining_f"]),
                                              stamp=int(r["stamp"]), burst=int(r["burst"]), expire_at=int(r["expire_at"]),
                                              status=int(r["status"])), 0)

    def global_take(self, role_mask=6):
        from gubernator_amd.rows import Rows
        take = [k for k, r in self.pending.items() if (role_mask >> r["role"]) & 1]
        rows = [self.pending.pop(k) for k in take]
        return Rows.from_dicts(rows) if rows else Rows.empty()

    def get_item(self, key, now_ms):
        return self.o.get_item(key, now_ms)
