"""ctypes binding of include/guber_wire.h: serialized GetRateLimitsReq / GetPeerRateLimitsReq payloads ->
one SoA device batch -> serialized responses (the step immediately before / after the hot path,
gubernator.proto:137-203, gubernator.go:189-220)."""
import ctypes as C

import numpy as np

from . import GuberError, lib
from .abi import GuberBatch, GuberResult

E_WIRE_MALFORMED, E_WIRE_TOO_LARGE, E_WIRE_FULL = -20, -21, -22
WIRE_SYMBOLS = [
    "guber_wire_batch_create", "guber_wire_batch_destroy", "guber_wire_batch_reset", "guber_wire_batch_size",
    "guber_wire_decode_requests", "guber_wire_batch_view", "guber_wire_batch_result", "guber_wire_batch_pre_errors",
    "guber_wire_eval", "guber_wire_encode_bound", "guber_wire_encode_responses",
    "guber_wire_items_create", "guber_wire_items_destroy", "guber_wire_decode_globals", "guber_wire_encode_globals",
    "guber_wire_dev_create", "guber_wire_dev_destroy", "guber_wire_dev_decode", "guber_wire_dev_buffer", "guber_wire_dev_decode_staged",
    "guber_wire_dev_eval", "guber_wire_dev_eval_front", "guber_wire_dev_columns",
    "guber_wire_dev_set_stream", "guber_wire_dev_decode_staged_async", "guber_wire_dev_decode_collect", "guber_wire_dev_eval_front_async",
    "guber_wire_dev_eval_collect", "guber_wire_dev_route_front_async", "guber_wire_dev_route_ready",
    "guber_wire_pool_create", "guber_wire_pool_destroy", "guber_wire_pool_get_rate_limits", "guber_wire_pool_response_bound",
    "guber_wire_pool_set_clock", "guber_wire_pool_stats",
]
_bound = False


def _lib():
    global _bound
    L = lib()
    if not _bound:
        L.guber_wire_batch_create.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.guber_wire_batch_destroy.argtypes = [C.c_void_p]
        L.guber_wire_batch_destroy.restype = None
        L.guber_wire_batch_reset.argtypes = [C.c_void_p, C.c_int64]
        L.guber_wire_batch_reset.restype = None
        L.guber_wire_batch_size.argtypes = [C.c_void_p]
        L.guber_wire_batch_size.restype = C.c_uint32
        L.guber_wire_decode_requests.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint8,
                                                 C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.guber_wire_batch_view.argtypes = [C.c_void_p]
        L.guber_wire_batch_view.restype = C.POINTER(GuberBatch)
        L.guber_wire_batch_result.argtypes = [C.c_void_p]
        L.guber_wire_batch_result.restype = C.POINTER(GuberResult)
        L.guber_wire_batch_pre_errors.argtypes = [C.c_void_p]
        L.guber_wire_batch_pre_errors.restype = C.POINTER(C.c_uint8)
        L.guber_wire_eval.argtypes = [C.c_void_p, C.c_void_p]
        L.guber_wire_encode_bound.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.guber_wire_encode_bound.restype = C.c_size_t
        L.guber_wire_encode_responses.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t,
                                                  C.POINTER(C.c_size_t)]
        L.guber_wire_items_create.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.guber_wire_items_destroy.argtypes = [C.c_void_p]
        L.guber_wire_items_destroy.restype = None
        L.guber_wire_decode_globals.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
        L.guber_wire_encode_globals.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(GuberResult), C.c_uint32,
                                                C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        _bound = True
    return L


class WireBatch:
    """One device batch under construction from any number of RPC payloads."""

    def __init__(self, max_items=65536, max_key_bytes=4 << 20, pinned=False):
        self.L = _lib()
        self.h = C.c_void_p()
        rc = self.L.guber_wire_batch_create(max_items, max_key_bytes, 1 if pinned else 0, C.byref(self.h))
        if rc:
            raise GuberError(rc, self.L.guber_strerror(rc).decode())

    def close(self):
        if self.h:
            self.L.guber_wire_batch_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, now_ms):
        self.L.guber_wire_batch_reset(self.h, now_ms)

    def __len__(self):
        return self.L.guber_wire_batch_size(self.h)

    def decode(self, payload: bytes, max_per_rpc=0, is_owner=True):
        """Append one serialized request message; returns (first, count).  Raises GuberError with the code
        E_WIRE_MALFORMED / E_WIRE_TOO_LARGE / E_WIRE_FULL (nothing appended)."""
        first, count = C.c_uint32(), C.c_uint32()
        rc = self.L.guber_wire_decode_requests(self.h, payload, len(payload), max_per_rpc, 1 if is_owner else 0,
                                               C.byref(first), C.byref(count))
        if rc:
            raise GuberError(rc, self.L.guber_strerror(rc).decode())
        return first.value, count.value

    def view(self):
        return self.L.guber_wire_batch_view(self.h).contents

    def result(self):
        return self.L.guber_wire_batch_result(self.h).contents

    def pre_errors(self):
        n = len(self)
        return np.ctypeslib.as_array(self.L.guber_wire_batch_pre_errors(self.h), shape=(n,)).copy() if n else np.zeros(0, np.uint8)

    def arrays(self):
        """numpy copies of the SoA (tests / debugging)."""
        v, n = self.view(), len(self)
        def arr(ptr, dt):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(dt)), shape=(n,)).copy() if n and ptr else None
        off = np.ctypeslib.as_array(C.cast(v.key_off, C.POINTER(C.c_uint32)), shape=(n + 1,)).copy()
        kb = C.string_at(v.key_bytes, int(off[-1])) if n else b""
        return dict(n=n, now_ms=v.now_ms, keys=[kb[off[i]:off[i + 1]] for i in range(n)],
                    hits=arr(v.hits, C.c_int64), limit=arr(v.limit, C.c_int64), duration=arr(v.duration, C.c_int64),
                    burst=arr(v.burst, C.c_int64), created_at=arr(v.created_at, C.c_int64),
                    algorithm=arr(v.algorithm, C.c_uint8), behavior=arr(v.behavior, C.c_uint32),
                    is_owner=arr(v.is_owner, C.c_uint8), greg_expire=arr(v.greg_expire, C.c_int64),
                    greg_duration=arr(v.greg_duration, C.c_int64))

    def eval(self, engine):
        """guber_eval_batch on the batch's own arrays (needs the GPU engine)."""
        rc = self.L.guber_wire_eval(engine.h, self.h)
        if rc:
            raise GuberError(rc, self.L.guber_strerror(rc).decode())

    def encode(self, first, count, wrap_errors=True):
        cap = self.L.guber_wire_encode_bound(self.h, first, count)
        buf = (C.c_uint8 * max(cap, 1))()
        n = C.c_size_t()
        rc = self.L.guber_wire_encode_responses(self.h, first, count, 1 if wrap_errors else 0, buf, cap, C.byref(n))
        if rc:
            raise GuberError(rc, self.L.guber_strerror(rc).decode())
        return bytes(buf[:n.value])


class WireItems:
    """UpdatePeerGlobalsReq payload -> the CacheItems the receiver installs (gubernator.go:425-459), ready for guber_add_items."""

    def __init__(self, max_items=65536, max_key_bytes=4 << 20):
        self.L = _lib()
        self.h = C.c_void_p()
        rc = self.L.guber_wire_items_create(max_items, max_key_bytes, C.byref(self.h))
        if rc:
            raise GuberError(rc, self.L.guber_strerror(rc).decode())

    def close(self):
        if self.h:
            self.L.guber_wire_items_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode(self, payload: bytes, now_ms):
        """-> (ctypes array view of GuberItem, count); valid until the next decode."""
        from .abi import GuberItem
        out, n = C.c_void_p(), C.c_uint32()
        rc = self.L.guber_wire_decode_globals(self.h, payload, len(payload), now_ms, C.byref(out), C.byref(n))
        if rc:
            raise GuberError(rc, self.L.guber_strerror(rc).decode())
        return C.cast(out, C.POINTER(GuberItem)), n.value


def encode_globals(host_batch, status):
    """UpdatePeerGlobalsReq bytes for the update rows in `host_batch` (HostBatch: keys, algorithm, duration, created_at) and
    their hits = 0 status (HostResult), as broadcastPeers builds it (global.go:234-262)."""
    import numpy as np
    L = _lib()
    n = host_batch.n
    created = host_batch.created_at if host_batch.created_at is not None else np.full(max(n, 1), host_batch.now_ms, np.int64)
    need = C.c_size_t()
    args = (host_batch.key_bytes.ctypes.data, host_batch.key_off.ctypes.data, host_batch.algorithm.ctypes.data,
            host_batch.duration.ctypes.data, created.ctypes.data, C.byref(status.c), n)
    L.guber_wire_encode_globals(*args, None, 0, C.byref(need))
    buf = (C.c_uint8 * max(need.value, 1))()
    rc = L.guber_wire_encode_globals(*args, buf, need.value, C.byref(need))
    if rc:
        raise GuberError(rc, L.guber_strerror(rc).decode())
    return bytes(buf[:need.value])


class _Columns(C.Structure):
    _fields_ = [("n", C.c_uint32), ("key_stride", C.c_uint32), ("key_rows", C.c_void_p), ("key_len", C.c_void_p),
                ("hits", C.c_void_p), ("limit", C.c_void_p), ("duration", C.c_void_p), ("burst", C.c_void_p), ("created_at", C.c_void_p),
                ("behavior", C.c_void_p), ("algo_raw", C.c_void_p), ("algorithm", C.c_void_p), ("is_owner", C.c_void_p), ("pre_err", C.c_void_p)]


class DevWireDecoder:
    """guber_wire_dev_*: serialized RPC payloads decoded ON THE DEVICE into one batch of the engine (guber_kernels_wire.h)."""

    def __init__(self, engine, max_items=65536, max_payload_bytes=8 << 20, max_rpcs=4096):
        L = self.L = _lib()
        L.guber_wire_dev_create.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.guber_wire_dev_destroy.argtypes = [C.c_void_p]
        L.guber_wire_dev_destroy.restype = None
        L.guber_wire_dev_decode.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int64,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        L.guber_wire_dev_buffer.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.guber_wire_dev_decode_staged.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int64,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        L.guber_wire_dev_eval.argtypes = [C.c_void_p, C.POINTER(GuberResult)]
        L.guber_wire_dev_columns.argtypes = [C.c_void_p, C.POINTER(_Columns)]
        self.engine = engine
        self.h = C.c_void_p()
        rc = L.guber_wire_dev_create(engine.h, max_items, max_payload_bytes, max_rpcs, C.byref(self.h))
        if rc:
            raise GuberError(rc, lib().guber_last_error().decode())

    def decode(self, payloads, now_ms, is_owner=None, max_per_rpc=1000):
        """-> (status[nrpc], first[nrpc], count[nrpc], n_items)"""
        n = len(payloads)
        msgs = (C.c_char_p * max(n, 1))(*payloads)
        lens = np.array([len(p) for p in payloads], np.uint32) if n else np.zeros(1, np.uint32)
        own = None if is_owner is None else np.asarray(is_owner, np.uint8)
        status, first, count = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32)
        items = C.c_uint32(0)
        rc = self.L.guber_wire_dev_decode(self.h, msgs, lens.ctypes.data, n, own.ctypes.data if own is not None else None, max_per_rpc, now_ms,
                                          status.ctypes.data, first.ctypes.data, count.ctypes.data, C.byref(items))
        if rc:
            raise GuberError(rc, lib().guber_last_error().decode())
        self.n = items.value
        return status[:n], first[:n], count[:n], items.value

    def buffer(self):
        """the decoder's pinned staging buffer as a writable numpy view (guber_wire_dev_buffer): what a receive path reads its sockets into"""
        p, cap = C.c_void_p(), C.c_size_t()
        rc = self.L.guber_wire_dev_buffer(self.h, C.byref(p), C.byref(cap))
        if rc:
            raise GuberError(rc, lib().guber_last_error().decode())
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (cap.value,))

    def decode_staged(self, offs, lens, now_ms, is_owner=None, max_per_rpc=1000):
        """payloads that already lie in buffer() at offs[] (16-byte aligned, ascending) -> (status[nrpc], first[nrpc], count[nrpc], n_items)"""
        offs, lens = np.ascontiguousarray(offs, np.uint32), np.ascontiguousarray(lens, np.uint32)
        n = len(offs)
        own = None if is_owner is None else np.asarray(is_owner, np.uint8)
        status, first, count = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32)
        items = C.c_uint32(0)
        rc = self.L.guber_wire_dev_decode_staged(self.h, offs.ctypes.data, lens.ctypes.data, n, own.ctypes.data if own is not None else None, max_per_rpc,
                                                 now_ms, status.ctypes.data, first.ctypes.data, count.ctypes.data, C.byref(items))
        if rc:
            raise GuberError(rc, lib().guber_last_error().decode())
        self.n = items.value
        return status[:n], first[:n], count[:n], items.value

    def eval(self):
        from .abi import HostResult
        res = HostResult(max(self.n, 1))
        rc = self.L.guber_wire_dev_eval(self.h, C.byref(res.c))
        if rc:
            raise GuberError(rc, lib().guber_last_error().decode())
        return res

    def eval_front(self, front):
        """the decoded batch through a front (guber_front_*): routed to the front's engines on the device, answered in the items' order"""
        from .abi import HostResult
        self.L.guber_wire_dev_eval_front.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(GuberResult)]
        res = HostResult(max(self.n, 1))
        rc = self.L.guber_wire_dev_eval_front(self.h, front.h, C.byref(res.c))
        if rc:
            raise GuberError(rc, lib().guber_last_error().decode())
        return res

    def columns(self):
        """the decoded columns as numpy arrays (+ keys: list of bytes)"""
        c = _Columns()
        rc = self.L.guber_wire_dev_columns(self.h, C.byref(c))
        if rc:
            raise GuberError(rc, lib().guber_last_error().decode())
        n = c.n
        def arr(ptr, dt):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), (n,)).copy() if n else np.zeros(0, dt)
        out = {k: arr(getattr(c, k), dt) for k, dt in (("hits", np.int64), ("limit", np.int64), ("duration", np.int64), ("burst", np.int64),
                                                       ("created_at", np.int64), ("key_len", np.uint32), ("behavior", np.uint32), ("algo_raw", np.int32),
                                                       ("algorithm", np.uint8), ("is_owner", np.uint8), ("pre_err", np.uint8))}
        rows = np.ctypeslib.as_array(C.cast(c.key_rows, C.POINTER(C.c_uint8)), (n * c.key_stride,)).reshape(n, c.key_stride) if n else np.zeros((0, 8), np.uint8)
        out["keys"] = [bytes(rows[i, :min(int(out["key_len"][i]), c.key_stride)]) for i in range(n)]
        return out

    def close(self):
        if getattr(self, "h", None):
            self.L.guber_wire_dev_destroy(self.h)
            self.h = None


class WirePoolConfig(C.Structure):
    _fields_ = [("stages", C.c_uint32), ("max_items", C.c_uint32), ("max_payload_bytes", C.c_uint32), ("max_rpcs", C.c_uint32),
                ("batch_wait_us", C.c_uint32), ("max_per_rpc", C.c_uint32), ("spin_us", C.c_uint32), ("decodes_queued", C.c_uint32)]


class WirePoolStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("rpcs", "items", "stages", "sealed_full", "sealed_wait", "sealed_idle", "open_waits",
                                           "fill_us_sum", "decode_us_sum", "eval_us_sum", "host_decode_ns", "host_route_ns", "host_eval_ns")]


class WirePool:
    """guber_wire_pool_* (include/guber_wire.h): the payload stage — V1Instance.GetRateLimits on the SERIALIZED messages
    (gubernator.go:183-306).  get_rate_limits() is what one gRPC handler thread calls; it may be called from many threads at once
    (the call releases the GIL)."""

    def __init__(self, engines, placement=None, **cfg):
        L = self.L = _lib()
        L.guber_wire_pool_create.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p, C.POINTER(WirePoolConfig), C.POINTER(C.c_void_p)]
        L.guber_wire_pool_destroy.argtypes = [C.c_void_p]
        L.guber_wire_pool_destroy.restype = None
        L.guber_wire_pool_get_rate_limits.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.guber_wire_pool_response_bound.argtypes = [C.c_char_p, C.c_size_t]
        L.guber_wire_pool_response_bound.restype = C.c_size_t
        L.guber_wire_pool_set_clock.argtypes = [C.c_void_p, C.c_int64]
        L.guber_wire_pool_stats.argtypes = [C.c_void_p, C.POINTER(WirePoolStats)]
        self.engines = list(engines)
        self.placement = placement
        hs = (C.c_void_p * len(self.engines))(*[e.h for e in self.engines])
        rule = placement.export() if placement is not None else None
        c = WirePoolConfig(**cfg)
        self.h = C.c_void_p()
        rc = L.guber_wire_pool_create(hs, len(self.engines), C.byref(rule) if rule is not None else None, C.byref(c), C.byref(self.h))
        if rc:
            raise GuberError(rc, lib().guber_last_error().decode())

    def set_clock(self, now_ms):
        self.L.guber_wire_pool_set_clock(self.h, now_ms)

    def get_rate_limits(self, payload, is_owner=True, wrap_errors=True, cap=None):
        """serialized GetRateLimitsReq -> serialized GetRateLimitsResp (GuberError for a message that is turned away whole)"""
        cap = self.L.guber_wire_pool_response_bound(payload, len(payload)) if cap is None else cap
        buf = C.create_string_buffer(max(cap, 1))
        n = C.c_size_t(0)
        rc = self.L.guber_wire_pool_get_rate_limits(self.h, payload, len(payload), 1 if is_owner else 0, 1 if wrap_errors else 0, buf, cap, C.byref(n))
        if rc:
            raise GuberError(rc, lib().guber_last_error().decode())
        return buf.raw[:n.value]

    def stats(self):
        st = WirePoolStats()
        self.L.guber_wire_pool_stats(self.h, C.byref(st))
        return {k: int(getattr(st, k)) for k, _ in WirePoolStats._fields_}

    def close(self):
        if getattr(self, "h", None):
            self.L.guber_wire_pool_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
