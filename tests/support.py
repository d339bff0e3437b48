"""Shared test plumbing: ctypes views of the C-ABI structs (include/guber_gpu.h), a batch
builder, and a wrapper around the CPU oracle (oracle/libguber_oracle.so).

The oracle is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg load it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TOKEN, LEAKY = 0, 1
UNDER, OVER = 0, 1
NO_BATCHING, GLOBAL, GREGORIAN, RESET_REMAINING, MULTI_REGION, DRAIN_OVER_LIMIT = 1, 2, 4, 8, 16, 32

ITEM_ERR_TEXT = {
    1: "Invalid rate limit algorithm",
    2: "`Duration = GregorianWeeks` not yet supported; consider making a PR!`",
    3: "behavior DURATION_IS_GREGORIAN is set; but `Duration` is not a valid gregorian interval",
}


from gubernator_amd.abi import (GuberConfig, GuberBatch, GuberResult, GuberItem, GuberStats, HostBatch,  # noqa: E402,F401
                                HostResult, assert_results_equal, make_item, item_dict, _ptr)


# ------------------------------------------------------------------------------------------------
# Store (store.go:49-65) test double: the reference's MockStore2 (store_test.go) in python
# ------------------------------------------------------------------------------------------------
class MockStore:
    """Records every call in order; `items` (key -> item dict) is what Get returns, like the mock's
    `.Return(storedItem, true)`.  on_change also keeps the last item per key (a write-through store)."""

    def __init__(self, items=None, write_through=False):
        self.items = dict(items or {})
        self.calls = []
        self.write_through = write_through

    def get(self, req_index, key):
        self.calls.append(("get", req_index, key))
        d = self.items.get(key)
        # "It's up to the store to expire old rate limit items" (store.go:52-53): a write-through store drops them
        now = getattr(self, "now", None)
        if d is not None and self.write_through and now is not None and \
                (d.get("expire_at", 0) < now or (d.get("invalid_at", 0) and d["invalid_at"] < now)):
            return None
        return d

    def on_change(self, req_index, key, item):
        self.calls.append(("on_change", req_index, key, item))
        if self.write_through:
            self.items[key] = dict(item, key=key)

    def remove(self, req_index, key):
        self.calls.append(("remove", req_index, key))
        if self.write_through:
            self.items.pop(key, None)

    def kinds(self):
        return [c[0] for c in self.calls]


_GET_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(GuberItem))
_CHG_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.POINTER(GuberItem))
_REM_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32)


class OracleStore(C.Structure):
    _fields_ = [("get", _GET_CB), ("on_change", _CHG_CB), ("remove", _REM_CB), ("user", C.c_void_p)]


def fill_item(out, d):
    """item dict -> *GuberItem (key left to the callee)"""
    out.algorithm = d["algorithm"]; out.status = d.get("status", 0); out.limit = d.get("limit", 0)
    out.duration = d.get("duration", 0); out.remaining = d.get("remaining", 0); out.remaining_f = d.get("remaining_f", 0.0)
    out.stamp = d.get("stamp", 0); out.burst = d.get("burst", 0); out.expire_at = d.get("expire_at", 0)
    out.invalid_at = d.get("invalid_at", 0)


def batch_key(batch, i):
    return bytes(batch.key_bytes[batch.key_off[i]:batch.key_off[i + 1]]).decode()


# ------------------------------------------------------------------------------------------------
# Oracle
# ------------------------------------------------------------------------------------------------
_ORACLE = None


def build_oracle():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


def oracle_lib():
    global _ORACLE
    if _ORACLE is None:
        path = os.path.join(ROOT, "oracle", "libguber_oracle.so")
        src = os.path.join(ROOT, "oracle", "guber_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build_oracle()
        lib = C.CDLL(path)
        lib.oracle_create.restype = C.c_void_p
        lib.oracle_create.argtypes = [C.c_uint64, C.c_uint32]
        lib.oracle_destroy.argtypes = [C.c_void_p]
        lib.oracle_eval_batch.argtypes = [C.c_void_p, C.POINTER(GuberBatch), C.POINTER(GuberResult)]
        lib.oracle_eval_batch_mt.argtypes = [C.c_void_p, C.POINTER(GuberBatch), C.POINTER(GuberResult), C.c_int]
        lib.oracle_eval_batch_store.argtypes = [C.c_void_p, C.POINTER(GuberBatch), C.POINTER(GuberResult), C.POINTER(OracleStore)]
        lib.oracle_add_item.argtypes = [C.c_void_p, C.POINTER(GuberItem), C.c_int64, C.POINTER(C.c_int)]
        lib.oracle_get_item.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_int64, C.POINTER(GuberItem),
                                        C.POINTER(C.c_int)]
        lib.oracle_remove_item.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
        lib.oracle_size.restype = C.c_int64
        lib.oracle_size.argtypes = [C.c_void_p]
        lib.oracle_each.restype = C.c_uint64
        lib.oracle_each.argtypes = [C.c_void_p, C.POINTER(GuberItem), C.c_uint64]
        lib.oracle_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        lib.oracle_worker_index_for_hash63.restype = C.c_uint32
        lib.oracle_worker_index_for_hash63.argtypes = [C.c_uint32, C.c_uint64]
        lib.oracle_xxhash64.restype = C.c_uint64
        lib.oracle_xxhash64.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
        lib.oracle_fnv1_64.restype = C.c_uint64
        lib.oracle_fnv1_64.argtypes = [C.c_char_p, C.c_size_t]
        lib.oracle_fnv1a_64.restype = C.c_uint64
        lib.oracle_fnv1a_64.argtypes = [C.c_char_p, C.c_size_t]
        lib.oracle_md5.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        lib.oracle_gregorian_expiration.argtypes = [C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
        lib.oracle_gregorian_duration.argtypes = [C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
        lib.oracle_ring_create.restype = C.c_void_p
        lib.oracle_ring_create.argtypes = [C.POINTER(C.c_char_p), C.c_uint32, C.c_uint32, C.c_int]
        lib.oracle_ring_destroy.argtypes = [C.c_void_p]
        lib.oracle_ring_get.restype = C.c_uint32
        lib.oracle_ring_get.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
        _ORACLE = lib
    return _ORACLE


class Oracle:
    """The reference's WorkerPool semantics on the CPU (sequential, per request)."""

    def __init__(self, cache_size=0, workers=1):
        self.lib = oracle_lib()
        self.h = self.lib.oracle_create(cache_size, workers)

    def close(self):
        if self.h:
            self.lib.oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def eval(self, batch, threads=0):
        res = HostResult(batch.n)
        if threads:
            self.lib.oracle_eval_batch_mt(self.h, C.byref(batch.c), C.byref(res.c), threads)
        else:
            self.lib.oracle_eval_batch(self.h, C.byref(batch.c), C.byref(res.c))
        return res

    def eval_store(self, batch, store):
        """The batch with Config.Store = `store` (a MockStore-like object), the reference's way: Get on a miss,
        OnChange / Remove from inside the algorithms."""
        res = HostResult(batch.n)

        def get(_u, i, out):
            d = store.get(i, batch_key(batch, i))
            if d is None:
                return 0
            fill_item(out.contents, d)
            return 1

        def chg(_u, i, item):
            store.on_change(i, batch_key(batch, i), item_dict(item.contents, key=batch_key(batch, i)))

        def rem(_u, i, key, klen):
            store.remove(i, C.string_at(key, klen).decode())
        cbs = OracleStore(_GET_CB(get), _CHG_CB(chg), _REM_CB(rem), None)
        self.lib.oracle_eval_batch_store(self.h, C.byref(batch.c), C.byref(res.c), C.byref(cbs))
        return res

    def add_item(self, item, now_ms=0):
        ex = C.c_int(0)
        self.lib.oracle_add_item(self.h, C.byref(item), now_ms, C.byref(ex))
        return bool(ex.value)

    def get_item(self, key, now_ms):
        kb = key if isinstance(key, bytes) else key.encode()
        out, found = GuberItem(), C.c_int(0)
        self.lib.oracle_get_item(self.h, kb, len(kb), now_ms, C.byref(out), C.byref(found))
        return item_dict(out, kb) if found.value else None

    def remove_item(self, key):
        kb = key if isinstance(key, bytes) else key.encode()
        self.lib.oracle_remove_item(self.h, kb, len(kb))

    def size(self):
        return self.lib.oracle_size(self.h)

    def each(self):
        n = self.lib.oracle_each(self.h, None, 0)
        arr = (GuberItem * max(n, 1))()
        self.lib.oracle_each(self.h, arr, n)
        return [item_dict(arr[i]) for i in range(n)]

    def counters(self):
        out = (C.c_uint64 * 4)()
        self.lib.oracle_counters(self.h, out)
        return tuple(out)


def gregorian(now_ms, d):
    """(greg_expire, greg_duration) as the host layer precomputes them (interval.go:84-148)."""
    lib = oracle_lib()
    e, g = C.c_int64(0), C.c_int64(0)
    rc = lib.oracle_gregorian_expiration(now_ms * 1_000_000, d, C.byref(e))
    rc2 = lib.oracle_gregorian_duration(now_ms * 1_000_000, d, C.byref(g))
    if rc != 0 or rc2 != 0:
        return 0, rc if rc != 0 else rc2
    return e.value, g.value
