"""ctypes binding of the native GLOBAL exchange (include/guber_gpu.h: guber_comm_* / guber_global_sync; implementation
gubernator_amd/csrc/guber_global_sync.h): N ranks of one process (one engine each, distinct GPUs or logical ranks on one GPU),
or one rank per process over RCCL."""
import ctypes as C

import numpy as np

from . import GuberError, abi, lib
from .abi import HostBatch

GLOBAL = abi.GLOBAL


class SyncStats(C.Structure):
    _fields_ = [("hits_rows_sent", C.c_uint64), ("hits_rows_applied", C.c_uint64), ("update_rows", C.c_uint64),
                ("items_installed", C.c_uint64), ("bytes_moved", C.c_uint64), ("fallbacks", C.c_uint64), ("ms", C.c_double)]

    def as_dict(self):
        return {f[0]: getattr(self, f[0]) for f in self._fields_}


_bound = False


def _lib():
    global _bound
    L = lib()
    if not _bound:
        L.guber_comm_create_local.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.guber_comm_unique_id.argtypes = [C.c_void_p]
        L.guber_comm_create_rank.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.guber_comm_destroy.argtypes = [C.c_void_p]
        L.guber_comm_destroy.restype = None
        L.guber_global_sync.argtypes = [C.c_void_p, C.c_int64, C.POINTER(SyncStats)]
        L.guber_comm_last_stats.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(SyncStats)]
        _bound = True
    return L


def _check(rc):
    if rc != 0:
        L = lib()
        raise GuberError(rc, f"{L.guber_strerror(rc).decode()} ({L.guber_last_error().decode()})")


def unique_id():
    buf = (C.c_uint8 * 128)()
    _check(_lib().guber_comm_unique_id(buf))
    return bytes(buf)


class Rank:
    """One rank's request side: a batch of GLOBAL requests arriving from clients (V1Instance.GetRateLimits,
    gubernator.go:247-270) — owned keys are evaluated as the owner, the others against the local replica."""

    def __init__(self, node, rank, ring):
        self.node, self.rank, self.ring = node, rank, ring

    def evaluate(self, keys, hits, limit, duration, now_ms, **kw):
        hb = HostBatch(keys, hits, limit, duration, now_ms, **kw)
        owner = self.ring.route((hb.key_bytes, hb.key_off)) if hb.n else np.zeros(0, np.uint32)
        kw = dict(kw)
        kw["behavior"] = np.broadcast_to(np.asarray(kw.get("behavior", 0), np.uint32), (hb.n,)) | np.uint32(GLOBAL)
        return self.node.eval(HostBatch((hb.key_bytes, hb.key_off), hits, limit, duration, now_ms,
                                        is_owner=(owner == self.rank).astype(np.uint8), **kw))


class Comm:
    """guber_comm_t.  Comm.local(engines, ring, use_rccl) = every rank in this process; Comm.rank(engine, rank, world, id, ring)
    = this process is one rank."""

    def __init__(self, handle, engines, first_rank, ring):
        self.h, self.engines, self.ring = handle, list(engines), ring
        self.ranks = [Rank(e, first_rank + i, ring) for i, e in enumerate(self.engines)]

    @staticmethod
    def local(engines, ring, use_rccl=False):
        arr = (C.c_void_p * len(engines))(*[e.h for e in engines])
        h = C.c_void_p()
        _check(_lib().guber_comm_create_local(arr, len(engines), ring.h if ring is not None else None, 1 if use_rccl else 0, C.byref(h)))
        return Comm(h, engines, 0, ring)

    @staticmethod
    def rank(engine, rank, world, uid, ring):
        h = C.c_void_p()
        buf = (C.c_uint8 * 128)(*uid) if uid else None
        _check(_lib().guber_comm_create_rank(engine.h, rank, world, buf, ring.h if ring is not None else None, C.byref(h)))
        return Comm(h, [engine], rank, ring)

    def sync(self, now_ms):
        """one GlobalSyncWait tick; -> per local rank {hits_sent, hits_applied, broadcast, installed}"""
        st = SyncStats()
        _check(_lib().guber_global_sync(self.h, now_ms, C.byref(st)))
        self.last = st.as_dict()
        out = []
        for i in range(len(self.engines)):
            r = SyncStats()
            _check(_lib().guber_comm_last_stats(self.h, i, C.byref(r)))
            out.append(dict(hits_sent=r.hits_rows_sent, hits_applied=r.hits_rows_applied, broadcast=r.update_rows,
                            installed=r.items_installed, bytes=r.bytes_moved))
        return out

    def close(self):
        if getattr(self, "h", None):
            _lib().guber_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass
