"""ctypes view of the C ABI (include/guber_gpu.h): struct layouts, SoA batch / result holders."""
import ctypes as C

import numpy as np

TOKEN, LEAKY = 0, 1
UNDER, OVER = 0, 1
NO_BATCHING, GLOBAL, GREGORIAN, RESET_REMAINING, MULTI_REGION, DRAIN_OVER_LIMIT = 1, 2, 4, 8, 16, 32

ITEM_OK, ITEM_E_INVALID_ALGORITHM, ITEM_E_GREGORIAN_WEEKS, ITEM_E_GREGORIAN_INVALID = 0, 1, 2, 3
ITEM_E_EMPTY_KEY, ITEM_E_RETRY, ITEM_E_TABLE_FULL, ITEM_E_KEY_TOO_LONG = 4, 5, 6, 7


class GuberConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("cache_size", C.c_uint64),
                ("table_slots", C.c_uint64), ("max_batch", C.c_uint32), ("max_key_bytes", C.c_uint32),
                ("stream", C.c_void_p), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class GuberBatch(C.Structure):
    _fields_ = [("n", C.c_uint32), ("reserved", C.c_uint32),
                ("key_bytes", C.c_void_p), ("key_off", C.c_void_p),
                ("hits", C.c_void_p), ("limit", C.c_void_p), ("duration", C.c_void_p),
                ("burst", C.c_void_p), ("created_at", C.c_void_p), ("algorithm", C.c_void_p),
                ("behavior", C.c_void_p), ("is_owner", C.c_void_p),
                ("greg_expire", C.c_void_p), ("greg_duration", C.c_void_p), ("now_ms", C.c_int64)]


class GuberResult(C.Structure):
    _fields_ = [("status", C.c_void_p), ("limit", C.c_void_p), ("remaining", C.c_void_p),
                ("reset_time", C.c_void_p), ("err", C.c_void_p),
                ("over_limit_count", C.c_uint64), ("cache_hits", C.c_uint64),
                ("cache_misses", C.c_uint64), ("unexpired_evictions", C.c_uint64),
                ("cache_size", C.c_int64)]


class GuberItem(C.Structure):
    _fields_ = [("algorithm", C.c_uint8), ("status", C.c_uint8), ("reserved0", C.c_uint16),
                ("key_len", C.c_uint32), ("key", C.c_void_p),
                ("limit", C.c_int64), ("duration", C.c_int64), ("remaining", C.c_int64),
                ("remaining_f", C.c_double), ("stamp", C.c_int64), ("burst", C.c_int64),
                ("expire_at", C.c_int64), ("invalid_at", C.c_int64)]


class GuberStats(C.Structure):
    _fields_ = [("over_limit_count", C.c_uint64), ("cache_hits", C.c_uint64), ("cache_misses", C.c_uint64),
                ("unexpired_evictions", C.c_uint64), ("cache_size", C.c_int64), ("table_slots", C.c_uint64),
                ("tags_used", C.c_uint64), ("batches", C.c_uint64), ("retries", C.c_uint64),
                ("compactions", C.c_uint64), ("small_batches", C.c_uint64), ("fused_batches", C.c_uint64),
                ("eviction_passes", C.c_uint64), ("tail_rebuilds", C.c_uint64), ("batch_cuts", C.c_uint64)]


class GuberGlobalRows(C.Structure):
    _fields_ = [("n", C.c_uint32), ("key_stride", C.c_uint32), ("key_bytes", C.c_void_p), ("key_len", C.c_void_p),
                ("hits", C.c_void_p), ("limit", C.c_void_p), ("duration", C.c_void_p), ("burst", C.c_void_p),
                ("created_at", C.c_void_p), ("behavior", C.c_void_p), ("algorithm", C.c_void_p), ("role", C.c_void_p)]


class GuberKernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_uint64), ("total_ms", C.c_double), ("units", C.c_uint64)]


def _ptr(a):
    return None if a is None else a.ctypes.data


class GuberStoreEvents(C.Structure):
    _fields_ = [("flags", C.c_void_p), ("items", C.c_void_p)]


class GuberStoreReq(C.Structure):
    _fields_ = [("key", C.c_void_p), ("key_len", C.c_uint32), ("name_len", C.c_uint32), ("hits", C.c_int64), ("limit", C.c_int64),
                ("duration", C.c_int64), ("burst", C.c_int64), ("created_at", C.c_int64), ("algorithm", C.c_int32),
                ("behavior", C.c_uint32)]


STORE_GET_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(GuberStoreReq), C.POINTER(GuberItem))
STORE_CHG_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(GuberStoreReq), C.POINTER(GuberItem))
STORE_REM_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_uint32)


class GuberStoreCallbacks(C.Structure):
    _fields_ = [("get", STORE_GET_CB), ("on_change", STORE_CHG_CB), ("remove", STORE_REM_CB), ("user", C.c_void_p)]


class HostBatch:
    """SoA batch in numpy arrays + the ctypes struct pointing at them."""

    def __init__(self, keys, hits, limit, duration, now_ms, burst=None, created_at=None, algorithm=None,
                 behavior=None, is_owner=None, greg_expire=None, greg_duration=None):
        n = len(keys)
        self.n = n
        if isinstance(keys, tuple):  # (key_bytes uint8 array, key_off uint32 array) prebuilt
            self.key_bytes, self.key_off = keys
            n = self.n = len(self.key_off) - 1
        else:
            bs = [k if isinstance(k, bytes) else k.encode() for k in keys]
            off = np.zeros(n + 1, np.uint32)
            if n:
                off[1:] = np.cumsum([len(b) for b in bs])
            self.key_off = off
            self.key_bytes = np.frombuffer(b"".join(bs) + b"\0" * 8, np.uint8).copy()   # (the kernels read keys in 8-byte words)

        def arr(x, dt):
            if x is None:
                return None
            a = np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=dt), (n,)))
            return a.copy() if not a.flags.writeable else a

        self.hits = arr(hits, np.int64)
        self.limit = arr(limit, np.int64)
        self.duration = arr(duration, np.int64)
        self.burst = arr(burst, np.int64)
        self.created_at = arr(created_at, np.int64)
        self.algorithm = arr(algorithm if algorithm is not None else 0, np.uint8)
        self.behavior = arr(behavior if behavior is not None else 0, np.uint32)
        self.is_owner = arr(is_owner, np.uint8)
        self.greg_expire = arr(greg_expire, np.int64)
        self.greg_duration = arr(greg_duration, np.int64)
        self.now_ms = int(now_ms)
        self.c = GuberBatch(n, 0, _ptr(self.key_bytes), _ptr(self.key_off), _ptr(self.hits), _ptr(self.limit),
                            _ptr(self.duration), _ptr(self.burst), _ptr(self.created_at), _ptr(self.algorithm),
                            _ptr(self.behavior), _ptr(self.is_owner), _ptr(self.greg_expire),
                            _ptr(self.greg_duration), self.now_ms)


class HostResult:
    def __init__(self, n):
        self.n = n
        m = max(n, 1)
        self.status = np.full(m, 0xEE, np.uint8)
        self.limit = np.full(m, -7777, np.int64)
        self.remaining = np.full(m, -7777, np.int64)
        self.reset_time = np.full(m, -7777, np.int64)
        self.err = np.full(m, 0xEE, np.uint8)
        self.c = GuberResult(_ptr(self.status), _ptr(self.limit), _ptr(self.remaining), _ptr(self.reset_time),
                             _ptr(self.err), 0, 0, 0, 0, 0)

    def rows(self):
        n = self.n
        return list(zip(self.status[:n].tolist(), self.limit[:n].tolist(), self.remaining[:n].tolist(),
                        self.reset_time[:n].tolist(), self.err[:n].tolist()))

    def arrays(self):
        n = self.n
        return (self.status[:n], self.limit[:n], self.remaining[:n], self.reset_time[:n], self.err[:n])

    def counters(self):
        return (self.c.over_limit_count, self.c.cache_hits, self.c.cache_misses, self.c.unexpired_evictions,
                self.c.cache_size)


def assert_results_equal(got, want, what=""):
    names = ("status", "limit", "remaining", "reset_time", "err")
    for name, g, w in zip(names, got.arrays(), want.arrays()):
        if not np.array_equal(g, w):
            bad = np.nonzero(g != w)[0]
            i = int(bad[0])
            raise AssertionError(
                f"{what}: {name} differs at {len(bad)} of {len(g)} positions; first idx {i}: "
                f"got {got.rows()[i]} want {want.rows()[i]}")


def make_item(key, algorithm, limit=0, duration=0, remaining=0, remaining_f=0.0, stamp=0, burst=0, expire_at=0,
              invalid_at=0, status=0):
    kb = key if isinstance(key, bytes) else key.encode()
    buf = C.create_string_buffer(kb, len(kb))
    it = GuberItem(algorithm, status, 0, len(kb), C.cast(buf, C.c_void_p), limit, duration, remaining,
                   remaining_f, stamp, burst, expire_at, invalid_at)
    it._keepalive = buf
    return it


def item_dict(it, key=None):
    d = dict(algorithm=it.algorithm, status=it.status, limit=it.limit, duration=it.duration,
             remaining=it.remaining, remaining_f=it.remaining_f, stamp=it.stamp, burst=it.burst,
             expire_at=it.expire_at, invalid_at=it.invalid_at)
    if key is not None:
        d["key"] = key
    elif it.key:
        d["key"] = C.string_at(it.key, it.key_len)
    return d


