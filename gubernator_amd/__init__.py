"""gubernator_amd — MI355X-native rate-limit evaluation engine (drop-in for gubernator's
WorkerPool.GetRateLimit path).  This package is a thin ctypes binding over the C ABI declared in
include/guber_gpu.h and implemented by gubernator_amd/libguber_hip.so (hand-written HIP for gfx950).

There is NO CPU fallback: importing works anywhere (so the ABI can be inspected), but creating an
Engine without the built library or without a GPU raises.
"""
import ctypes as C
import os

import numpy as np

from . import abi
from .abi import (GuberBatch, GuberConfig, GuberItem, GuberResult, GuberStats, HostBatch, HostResult,  # noqa: F401
                  item_dict, make_item)

_HERE = os.path.dirname(os.path.abspath(__file__))
# GUBER_HIP_LIB selects another build of the same library (the phase-timing measurement build, `make timing`)
LIB_PATH = os.environ.get("GUBER_HIP_LIB") or os.path.join(_HERE, "libguber_hip.so")

# every symbol include/guber_gpu.h declares
ABI_SYMBOLS = [
    "guber_engine_create", "guber_engine_destroy", "guber_eval_batch", "guber_eval_batch_dev", "guber_add_items",
    "guber_get_item", "guber_remove_item", "guber_size", "guber_dump", "guber_stats", "guber_synchronize",
    "guber_alloc_pinned", "guber_free_pinned", "guber_ring_create", "guber_ring_destroy", "guber_ring_route",
    "guber_ring_route_dev", "guber_ring_points", "guber_gregorian_expiration", "guber_gregorian_duration",
    "guber_xxhash64", "guber_fnv1_64", "guber_fnv1a_64", "guber_strerror", "guber_item_strerror",
    "guber_last_error", "guber_version", "guber_profile_enable", "guber_profile_read", "guber_profile_passes", "guber_set_timezone", "guber_global_take",
    "guber_pool_create", "guber_pool_destroy", "guber_pool_set_clock", "guber_pool_engine", "guber_pool_batches",
    "guber_pool_get_rate_limits", "guber_compact", "guber_probe_missing", "guber_eval_batch_store",
    "guber_eval_batches_dev", "guber_eval_batches_routed_dev", "guber_front_create", "guber_front_destroy", "guber_front_set_rule", "guber_front_eval_dev",
               "guber_front_synchronize", "guber_front_stream", "guber_front_stats", "guber_front_latencies", "guber_set_clock", "guber_comm_create_local", "guber_comm_unique_id", "guber_comm_create_rank",
    "guber_comm_destroy", "guber_global_sync", "guber_comm_last_stats", "guber_stage_create", "guber_stage_destroy",
    "guber_stage_batch", "guber_stage_result", "guber_stage_capacity", "guber_stage_submit", "guber_stage_wait", "guber_pool_create_multi",
    "guber_pool_shards", "guber_pool_device_of", "guber_pool_engine_at", "guber_pool_metrics", "guber_pool_set_store", "guber_pool_create_sharded", "guber_pool_shard_of", "guber_pool_load", "guber_pool_store", "guber_global_pending", "guber_global_take_dev", "guber_ring_route_rows_dev", "guber_add_items_dev",
    "guber_placement_create", "guber_placement_destroy", "guber_placement_shard", "guber_placement_version", "guber_placement_route_keys",
    "guber_placement_observe", "guber_placement_observe_keys", "guber_placement_rebalance", "guber_placement_info",
    "guber_stages_submit", "guber_stage_poll", "guber_stage_dest", "guber_stage_submit_routed", "guber_placement_plan", "guber_placement_commit", "guber_placement_cancel", "guber_placement_export", "guber_stage_route", "guber_stage_route_poll", "guber_move_items_by_hash", "guber_engine_stream", "guber_pool_global_engine", "guber_pool_global_sync", "guber_pool_rebalance", "guber_pool_get_rate_limits_owner", "guber_pool_add_item", "guber_pool_add_item_for", "guber_pool_load_hinted", "guber_pool_get_item", "guber_pool_size",
]

FLAG_TEST_WEAK_HASH, FLAG_TEST_FORCE_RADIX, FLAG_TEST_CAREFUL, FLAG_GLOBAL, FLAG_DIR_CLAIMS, FLAG_TEST_NO_SMALL = 1, 2, 4, 8, 16, 32
FLAG_TEST_FORCE_PART, FLAG_NO_PART = 64, 128

_lib = None


E_INVALID_ARG, E_NO_DEVICE, E_HIP, E_BATCH_TOO_LARGE, E_TABLE_FULL, E_NOMEM, E_KEY_TOO_LONG, E_NOT_FOUND = -1, -2, -3, -4, -5, -6, -7, -8   # include/guber_gpu.h


class GuberError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"guber error {code}: {msg}")
        self.code = code


def lib():
    """Load libguber_hip.so (built by __graft_entry__.build() / gubernator_amd/csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GuberError(-2, f"{LIB_PATH} is missing: build it with `make -C gubernator_amd/csrc` "
                                 "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.guber_engine_create.argtypes = [C.POINTER(GuberConfig), C.POINTER(C.c_void_p)]
        L.guber_engine_destroy.argtypes = [C.c_void_p]
        L.guber_engine_destroy.restype = None
        for name in ("guber_eval_batch", "guber_eval_batch_dev"):
            getattr(L, name).argtypes = [C.c_void_p, C.POINTER(GuberBatch), C.POINTER(GuberResult)]
        L.guber_eval_batches_dev.argtypes = [C.c_void_p, C.POINTER(GuberBatch), C.POINTER(GuberResult), C.c_uint32, C.POINTER(C.c_uint32)]
        L.guber_eval_batches_routed_dev.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(GuberBatch),
                                                    C.POINTER(GuberResult), C.c_uint32, C.POINTER(C.c_uint32)]
        L.guber_add_items.argtypes = [C.c_void_p, C.POINTER(GuberItem), C.c_uint32, C.c_void_p]
        L.guber_get_item.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_int64, C.POINTER(GuberItem),
                                     C.POINTER(C.c_int)]
        L.guber_remove_item.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
        L.guber_size.restype = C.c_int64
        L.guber_size.argtypes = [C.c_void_p]
        L.guber_dump.argtypes = [C.c_void_p, C.POINTER(GuberItem), C.c_uint64, C.c_void_p, C.c_uint64,
                                 C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.guber_stats.argtypes = [C.c_void_p, C.POINTER(GuberStats)]
        L.guber_synchronize.argtypes = [C.c_void_p]
        L.guber_compact.argtypes = [C.c_void_p, C.c_int64]
        L.guber_set_clock.argtypes = [C.c_void_p, C.c_int64]
        L.guber_probe_missing.argtypes = [C.c_void_p, C.POINTER(GuberBatch), C.c_void_p]
        L.guber_eval_batch_store.argtypes = [C.c_void_p, C.POINTER(GuberBatch), C.POINTER(GuberResult), C.POINTER(abi.GuberStoreEvents)]
        L.guber_global_take.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(abi.GuberGlobalRows)]
        L.guber_pool_create.argtypes = [C.POINTER(GuberConfig), C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.guber_pool_destroy.argtypes = [C.c_void_p]
        L.guber_pool_destroy.restype = None
        L.guber_pool_set_store.argtypes = [C.c_void_p, C.c_void_p]
        L.guber_pool_set_store.restype = None
        L.guber_pool_set_clock.argtypes = [C.c_void_p, C.c_int64]
        L.guber_pool_set_clock.restype = None
        L.guber_pool_engine.argtypes = [C.c_void_p]
        L.guber_pool_engine.restype = C.c_void_p
        L.guber_pool_batches.argtypes = [C.c_void_p]
        L.guber_pool_batches.restype = C.c_uint64
        L.guber_pool_get_rate_limits.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 11 + [C.POINTER(GuberResult), C.c_void_p, C.c_uint32]
        L.guber_pool_get_rate_limits_owner.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 12 + [C.POINTER(GuberResult), C.c_void_p, C.c_uint32]
        L.guber_profile_enable.argtypes = [C.c_void_p, C.c_int]
        L.guber_profile_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.guber_profile_passes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.guber_alloc_pinned.restype = C.c_void_p
        L.guber_alloc_pinned.argtypes = [C.c_size_t]
        L.guber_free_pinned.argtypes = [C.c_void_p]
        L.guber_free_pinned.restype = None
        L.guber_ring_create.argtypes = [C.POINTER(C.c_char_p), C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
        L.guber_ring_destroy.argtypes = [C.c_void_p]
        L.guber_ring_destroy.restype = None
        L.guber_ring_route.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.guber_ring_route_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.guber_ring_points.restype = C.c_uint32
        L.guber_ring_points.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.guber_gregorian_expiration.argtypes = [C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
        L.guber_gregorian_duration.argtypes = [C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
        L.guber_xxhash64.restype = C.c_uint64
        L.guber_xxhash64.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
        L.guber_fnv1_64.restype = C.c_uint64
        L.guber_fnv1_64.argtypes = [C.c_char_p, C.c_size_t]
        L.guber_fnv1a_64.restype = C.c_uint64
        L.guber_fnv1a_64.argtypes = [C.c_char_p, C.c_size_t]
        for name in ("guber_strerror", "guber_item_strerror", "guber_last_error", "guber_version"):
            getattr(L, name).restype = C.c_char_p
        L.guber_strerror.argtypes = [C.c_int]
        L.guber_item_strerror.argtypes = [C.c_uint8]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        L = lib()
        raise GuberError(rc, f"{L.guber_strerror(rc).decode()} ({L.guber_last_error().decode()})")


def gregorian(now_ms, d):
    """(greg_expire, greg_duration) for a DURATION_IS_GREGORIAN request (interval.go:84-148);
    greg_duration < 0 carries the reference's error."""
    L = lib()
    e, g = C.c_int64(0), C.c_int64(0)
    rc = L.guber_gregorian_expiration(now_ms * 1_000_000, d, C.byref(e))
    rc2 = L.guber_gregorian_duration(now_ms * 1_000_000, d, C.byref(g))
    if rc != 0 or rc2 != 0:
        return 0, rc if rc != 0 else rc2
    return e.value, g.value


class Ring:
    """ReplicatedConsistentHash (replicated_hash.go): which peer / GPU owns a key."""

    def __init__(self, peers, replicas=512, hash_kind="fnv1"):
        self.peers = list(peers)
        arr = (C.c_char_p * len(self.peers))(*[p.encode() for p in self.peers])
        self.h = C.c_void_p()
        _check(lib().guber_ring_create(arr, len(self.peers), replicas, 1 if hash_kind == "fnv1a" else 0,
                                       C.byref(self.h)))

    def route(self, keys):
        if isinstance(keys, tuple):
            kb, ko = keys
        else:
            hb = HostBatch(keys, 0, 0, 0, 0)
            kb, ko = hb.key_bytes, hb.key_off
        n = len(ko) - 1
        owner = np.zeros(max(n, 1), np.uint32)
        _check(lib().guber_ring_route(self.h, kb.ctypes.data, ko.ctypes.data, n, owner.ctypes.data))
        return owner[:n]

    def points(self):
        n = lib().guber_ring_points(self.h, None, None, 0)
        hh, oo = np.zeros(n, np.uint64), np.zeros(n, np.uint32)
        lib().guber_ring_points(self.h, hh.ctypes.data, oo.ctypes.data, n)
        return hh, oo

    def close(self):
        if self.h:
            lib().guber_ring_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RouteRule(C.Structure):
    """guber_route_rule_t (include/guber_gpu.h): the placement in the form the device applies it (guber_stage_route)"""
    _fields_ = [("n_shards", C.c_uint32), ("per", C.c_uint32), ("step", C.c_uint64), ("inv_step", C.c_uint64), ("inv_sub", C.c_uint64),
                ("table", C.c_void_p), ("ex_cells", C.c_uint32), ("ex_n", C.c_uint32), ("ex_hash", C.c_void_p), ("ex_shard", C.c_void_p),
                ("global_engine", C.c_int32)]


class Placement:
    """guber_placement_t: which logical shard of a GPU holds a key — hash slots -> shards plus individually placed hot keys,
    fitted to observed traffic (include/guber_gpu.h; the generalisation of WorkerPool.getWorker, workers.go:180-184)."""

    def __init__(self, n_shards, n_slots=0):
        L = lib()
        L.guber_placement_create.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.guber_placement_destroy.argtypes = [C.c_void_p]
        L.guber_placement_destroy.restype = None
        L.guber_placement_shard.argtypes = [C.c_void_p, C.c_uint64]
        L.guber_placement_shard.restype = C.c_uint32
        L.guber_placement_route_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.guber_placement_observe_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.guber_placement_rebalance.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.guber_placement_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        self.n_shards = n_shards
        self.h = C.c_void_p()
        _check(L.guber_placement_create(n_shards, n_slots, C.byref(self.h)))

    def route_keys(self, key_bytes, key_off):
        """(shard uint32[n], hash uint64[n]) of packed keys"""
        n = len(key_off) - 1
        sh, hh = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint64)
        _check(lib().guber_placement_route_keys(self.h, key_bytes.ctypes.data, key_off.ctypes.data, n, sh.ctypes.data, hh.ctypes.data))
        return sh[:n], hh[:n]

    def shard(self, key_hash):
        return lib().guber_placement_shard(self.h, int(key_hash))

    def observe_keys(self, key_bytes, key_off):
        _check(lib().guber_placement_observe_keys(self.h, key_bytes.ctypes.data, key_off.ctypes.data, len(key_off) - 1))

    def rebalance(self, heavy_fraction=0.125, move_slots=True):
        """-> list of (key_hash, from, to) for the hot keys whose shard changed"""
        mv = np.zeros(64 * 4, np.uint32)          # guber_placement_move_t = u64 + 2 x u32
        nm = C.c_uint32(0)
        _check(lib().guber_placement_rebalance(self.h, heavy_fraction, 1 if move_slots else 0, mv.ctypes.data, 64, C.byref(nm)))
        rec = mv.view(np.dtype([("h", np.uint64), ("from", np.uint32), ("to", np.uint32)]))
        return [(int(r["h"]), int(r["from"]), int(r["to"])) for r in rec[:nm.value]]

    def export(self, global_engine=-1):
        """guber_placement_export -> RouteRule (valid while this placement lives)"""
        L = lib()
        L.guber_placement_export.argtypes = [C.c_void_p, C.POINTER(RouteRule)]
        r = RouteRule()
        _check(L.guber_placement_export(self.h, C.byref(r)))
        r.global_engine = global_engine
        return r

    def n_hot(self):
        nh = C.c_uint32(0)
        _check(lib().guber_placement_info(self.h, None, None, C.byref(nh)))
        return nh.value

    def close(self):
        if getattr(self, "h", None):
            lib().guber_placement_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One GPU-resident bucket table + the batched evaluation kernels (C ABI guber_engine_*)."""

    def __init__(self, cache_size=50_000, device=0, max_batch=65536, table_slots=0, max_key_bytes=0, stream=None,
                 flags=0):
        cfg = GuberConfig(C.sizeof(GuberConfig), device, cache_size, table_slots, max_batch, max_key_bytes,
                          stream, flags, 0)
        self.h = C.c_void_p()
        _check(lib().guber_engine_create(C.byref(cfg), C.byref(self.h)))
        self.max_batch = max_batch

    def close(self):
        if getattr(self, "h", None):
            lib().guber_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- hot path -------------------------------------------------------------------------------
    def eval(self, batch):
        """HostBatch -> HostResult through guber_eval_batch (host pointers, synchronous)."""
        res = HostResult(batch.n)
        _check(lib().guber_eval_batch(self.h, C.byref(batch.c), C.byref(res.c)))
        return res

    def probe_missing(self, batch):
        """missing[i] = 1: the key of request i is not resident at batch.now_ms (Store.Get is due, algorithms.go:45-51)."""
        out = np.zeros(max(batch.n, 1), np.uint8)
        _check(lib().guber_probe_missing(self.h, C.byref(batch.c), out.ctypes.data))
        return out[:batch.n]

    def eval_store_raw(self, batch):
        """guber_eval_batch_store: (HostResult, flags uint8[n], items GuberItem[n])."""
        res = HostResult(batch.n)
        flags = np.zeros(max(batch.n, 1), np.uint8)
        items = (GuberItem * max(batch.n, 1))()
        ev = abi.GuberStoreEvents(flags.ctypes.data, C.cast(items, C.c_void_p))
        _check(lib().guber_eval_batch_store(self.h, C.byref(batch.c), C.byref(res.c), C.byref(ev)))
        return res, flags[:batch.n], items

    def eval_store(self, batch, store):
        """One batch with a persistent Store configured, the way the Go shim drives it (include/guber_gpu.h,
        Config.Store section): Store.Get for the first request of every non-resident key -> guber_add_items ->
        guber_eval_batch_store -> Remove / OnChange callbacks in request order.
        `store` has get(req_index, key) -> item dict | None, on_change(req_index, key, item dict), remove(req_index, key)."""
        keys = [bytes(batch.key_bytes[batch.key_off[i]:batch.key_off[i + 1]]).decode() for i in range(batch.n)]
        asked = set()
        for i in np.nonzero(self.probe_missing(batch))[0].tolist():
            if keys[i] in asked or not keys[i]:
                continue
            asked.add(keys[i])
            d = store.get(i, keys[i])
            if d is not None:
                self.add_item(make_item(keys[i], d["algorithm"], limit=d.get("limit", 0), duration=d.get("duration", 0),
                                        remaining=d.get("remaining", 0), remaining_f=d.get("remaining_f", 0.0),
                                        stamp=d.get("stamp", 0), burst=d.get("burst", 0), expire_at=d.get("expire_at", 0),
                                        invalid_at=d.get("invalid_at", 0), status=d.get("status", 0)), batch.now_ms)
        res, flags, items = self.eval_store_raw(batch)
        for i in range(batch.n):
            if flags[i] & 2:
                store.remove(i, keys[i])
            if flags[i] & 1:
                store.on_change(i, keys[i], item_dict(items[i], key=keys[i]))
        return res

    def eval_dev(self, batch_struct, result_struct):
        """GuberBatch / GuberResult whose pointers are DEVICE pointers; asynchronous."""
        _check(lib().guber_eval_batch_dev(self.h, C.byref(batch_struct), C.byref(result_struct)))

    def eval_many_dev(self, batch_array, result_array, count):
        """ctypes arrays of GuberBatch / GuberResult (device pointers): enqueue `count` batches back to back."""
        _check(lib().guber_eval_batches_dev(self.h, batch_array, result_array, count, None))

    @staticmethod
    def eval_routed_dev(engines, which_array, batch_array, result_array, count):
        """one dispatcher for several engines: batch k -> engines[which[k]] (ctypes arrays, device pointers)"""
        hs = (C.c_void_p * len(engines))(*[e.h for e in engines])
        _check(lib().guber_eval_batches_routed_dev(hs, len(engines), which_array, batch_array, result_array, count, None))

    # -- cache operations -----------------------------------------------------------------------
    def add_item(self, item, now_ms=0):
        if now_ms:
            lib().guber_set_clock(self.h, now_ms)       # the clock Add's eviction classifies expired items against
        ex = (C.c_uint8 * 1)()
        _check(lib().guber_add_items(self.h, C.byref(item), 1, ex))
        return bool(ex[0])

    def add_items(self, items):
        arr = (GuberItem * len(items))(*items)
        ex = (C.c_uint8 * max(len(items), 1))()
        _check(lib().guber_add_items(self.h, arr, len(items), ex))
        return [bool(x) for x in ex[:len(items)]]

    def get_item(self, key, now_ms):
        kb = key if isinstance(key, bytes) else key.encode()
        out, found = GuberItem(), C.c_int(0)
        _check(lib().guber_get_item(self.h, kb, len(kb), now_ms, C.byref(out), C.byref(found)))
        return item_dict(out, kb) if found.value else None

    def remove_item(self, key):
        kb = key if isinstance(key, bytes) else key.encode()
        _check(lib().guber_remove_item(self.h, kb, len(kb)))

    def size(self):
        return lib().guber_size(self.h)

    def each(self):
        n, a = C.c_uint64(0), C.c_uint64(0)
        rc = lib().guber_dump(self.h, None, 0, None, 0, C.byref(n), C.byref(a))
        if rc not in (0, -6):
            _check(rc)
        cap, acap = n.value + 16, a.value + 1024
        items = (GuberItem * cap)()
        arena = C.create_string_buffer(acap)
        _check(lib().guber_dump(self.h, items, cap, arena, acap, C.byref(n), C.byref(a)))
        return [item_dict(items[i]) for i in range(n.value)]

    def counters(self):
        """(over_limit, cache_hits, cache_misses, unexpired_evictions, size) — the tuple HostResult.counters() has"""
        st = self.stats()
        return (st["over_limit_count"], st["cache_hits"], st["cache_misses"], st["unexpired_evictions"], st["cache_size"])

    def stats(self):
        s = GuberStats()
        _check(lib().guber_stats(self.h, C.byref(s)))
        return {f[0]: getattr(s, f[0]) for f in GuberStats._fields_}

    def global_take(self, role_mask=6):
        """Pending GLOBAL rows of the requested roles (bit 1 = hits for the owner, bit 2 = owner updates) as
        a rows.Rows (structure of numpy arrays); those queues are cleared."""
        from .rows import Rows
        rows = abi.GuberGlobalRows()
        _check(lib().guber_global_take(self.h, role_mask, C.byref(rows)))
        n = rows.n
        if n == 0:
            return Rows.empty(rows.key_stride or 8)
        def arr(ptr, dt):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(dt)), shape=(n,)).copy()
        km = np.ctypeslib.as_array(C.cast(rows.key_bytes, C.POINTER(C.c_uint8)), shape=(n, rows.key_stride)).copy()
        return Rows(km, arr(rows.key_len, C.c_uint32), arr(rows.hits, C.c_int64), arr(rows.limit, C.c_int64),
                    arr(rows.duration, C.c_int64), arr(rows.burst, C.c_int64), arr(rows.created_at, C.c_int64),
                    arr(rows.behavior, C.c_uint32), arr(rows.algorithm, C.c_uint8), arr(rows.role, C.c_uint8))

    def add_items_struct(self, items, keepalive=None):
        """guber_add_items on a numpy structured array laid out as guber_item_t (rows.ITEM_DTYPE)."""
        n = len(items)
        if n:
            _check(lib().guber_add_items(self.h, C.cast(items.ctypes.data, C.POINTER(GuberItem)), n, None))

    def profile(self, enable):
        _check(lib().guber_profile_enable(self.h, 1 if enable else 0))

    def profile_read(self):
        """{kernel name: (launches, total_ms)} since the last read."""
        cap = 32
        arr = (abi.GuberKernelTime * cap)()
        n = C.c_uint32(0)
        _check(lib().guber_profile_read(self.h, arr, cap, C.byref(n)))
        got = min(n.value, cap)                                       # (n = the kernels the library knows; it fills at most cap)
        self.last_profile_units = {arr[i].name.decode(): arr[i].units for i in range(got)}
        return {arr[i].name.decode(): (arr[i].launches, arr[i].total_ms) for i in range(got)}

    def profile_passes(self):
        """after profile_read(): microseconds every pipeline pass (one batch, or one fused group) took from its first kernel's start
        to its last kernel's end"""
        n = C.c_uint32(0)
        _check(lib().guber_profile_passes(self.h, None, 0, C.byref(n)))
        arr = (C.c_float * max(n.value, 1))()
        _check(lib().guber_profile_passes(self.h, arr, n.value, C.byref(n)))
        return [arr[i] for i in range(n.value)]

    def compact(self, now_ms):
        _check(lib().guber_compact(self.h, now_ms))

    def set_clock(self, now_ms):
        """clock.Freeze / clock.Advance for maintenance between batches (eviction after Add)"""
        _check(lib().guber_set_clock(self.h, now_ms))

    def synchronize(self):
        _check(lib().guber_synchronize(self.h))

    def stream_handle(self):
        """the hipStream_t the engine enqueues on (guber_engine_stream): hand it to further engines so that they share launches"""
        L = lib()
        L.guber_engine_stream.argtypes = [C.c_void_p]
        L.guber_engine_stream.restype = C.c_void_p
        return L.guber_engine_stream(self.h)

    def move_items_to(self, other, key_hashes):
        """guber_move_items_by_hash: the buckets of the keys with these XXH64 hashes leave this engine's table for `other`'s -> moved"""
        L = lib()
        L.guber_move_items_by_hash.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint32)]
        arr = (C.c_uint64 * max(len(key_hashes), 1))(*[int(h) for h in key_hashes])
        moved = C.c_uint32(0)
        _check(L.guber_move_items_by_hash(self.h, other.h, arr, len(key_hashes), C.byref(moved)))
        return moved.value

    def route_dev(self, ring, key_bytes_ptr, key_off_ptr, n, owner_ptr):
        _check(lib().guber_ring_route_dev(self.h, ring.h, key_bytes_ptr, key_off_ptr, n, owner_ptr))


class FrontStats(C.Structure):
    _fields_ = [("generations", C.c_uint64), ("forced_flushes", C.c_uint64), ("host_waits", C.c_uint64), ("host_wait_us", C.c_uint64)]


class Front:
    """guber_front_t (include/guber_gpu.h): generations of requests in ARRIVAL order, resident in HBM -> routed to the engines on the
    device (XXH64 + the placement's rule: WorkerPool.getWorker, workers.go:180-184) -> evaluated -> answered in ARRIVAL order
    (gubernator.proto:51-54)."""

    def __init__(self, engines, placement=None, max_n=65536, depth=0, global_engine=-1):
        L = lib()
        L.guber_front_create.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.guber_front_destroy.argtypes = [C.c_void_p]
        L.guber_front_destroy.restype = None
        L.guber_front_eval_dev.argtypes = [C.c_void_p, C.POINTER(GuberBatch), C.POINTER(GuberResult), C.c_uint32, C.POINTER(C.c_uint32)]
        L.guber_front_synchronize.argtypes = [C.c_void_p]
        L.guber_front_set_rule.argtypes = [C.c_void_p, C.c_void_p]
        L.guber_front_stats.argtypes = [C.c_void_p, C.POINTER(FrontStats)]
        L.guber_front_stream.argtypes = [C.c_void_p]
        L.guber_front_stream.restype = C.c_void_p
        self.engines = list(engines)
        self.placement = placement
        hs = (C.c_void_p * len(self.engines))(*[e.h for e in self.engines])
        self.h = C.c_void_p()
        rule = placement.export(global_engine=global_engine) if placement is not None else None   # (global_engine: index of the engine that takes Behavior_GLOBAL requests)
        _check(L.guber_front_create(hs, len(self.engines), C.byref(rule) if rule is not None else None, max_n, depth, C.byref(self.h)))

    def eval_dev(self, gen_array, result_array, count):
        """ctypes arrays of GuberBatch / GuberResult (device pointers, arrival order); asynchronous"""
        done = C.c_uint32(0)
        _check(lib().guber_front_eval_dev(self.h, gen_array, result_array, count, C.byref(done)))
        return done.value

    def synchronize(self):
        _check(lib().guber_front_synchronize(self.h))

    def stream_handle(self):
        return lib().guber_front_stream(self.h)

    def latencies(self):
        """microseconds per generation since the last call (profiling on the first engine)"""
        L = lib()
        L.guber_front_latencies.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        buf = np.zeros(1 << 16, np.float32)
        n = C.c_uint32(0)
        _check(L.guber_front_latencies(self.h, buf.ctypes.data, len(buf), C.byref(n)))
        return buf[:min(n.value, len(buf))].tolist()

    def stats(self):
        st = FrontStats()
        _check(lib().guber_front_stats(self.h, C.byref(st)))
        return {k: int(getattr(st, k)) for k, _ in FrontStats._fields_}

    def close(self):
        if getattr(self, "h", None):
            lib().guber_front_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Stage:
    """guber_stage_t: one batch's request / response arrays in device-visible host memory, filled in place (numpy views),
    submitted asynchronously.  Two stages per engine = the overlapped end-to-end path."""

    def __init__(self, engine, max_n, key_bytes_cap=0):
        L = lib()
        L.guber_stage_create.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.guber_stage_destroy.argtypes = [C.c_void_p]
        L.guber_stage_destroy.restype = None
        L.guber_stage_batch.argtypes = [C.c_void_p]
        L.guber_stage_batch.restype = C.POINTER(GuberBatch)
        L.guber_stage_result.argtypes = [C.c_void_p]
        L.guber_stage_result.restype = C.POINTER(GuberResult)
        L.guber_stage_capacity.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.guber_stage_capacity.restype = C.c_uint32
        L.guber_stage_submit.argtypes = [C.c_void_p]
        L.guber_stage_wait.argtypes = [C.c_void_p]
        self.h = C.c_void_p()
        self.engine = engine
        _check(L.guber_stage_create(engine.h, max_n, key_bytes_cap, C.byref(self.h)))
        kc = C.c_uint32()
        self.max_n = L.guber_stage_capacity(self.h, C.byref(kc))
        self.key_cap = kc.value
        self.b = L.guber_stage_batch(self.h).contents
        self.r = L.guber_stage_result(self.h).contents
        n = self.max_n

        def view(ptr, ct, count):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(count,))
        self.key_bytes = view(self.b.key_bytes, C.c_uint8, self.key_cap + 16)
        self.key_off = view(self.b.key_off, C.c_uint32, n + 1)
        self.hits, self.limit, self.duration, self.burst, self.created_at = (view(getattr(self.b, f), C.c_int64, n)
                                                                              for f in ("hits", "limit", "duration", "burst", "created_at"))
        self.algorithm = view(self.b.algorithm, C.c_uint8, n)
        self.behavior = view(self.b.behavior, C.c_uint32, n)
        self.is_owner = view(self.b.is_owner, C.c_uint8, n)
        self.status, self.err = view(self.r.status, C.c_uint8, n), view(self.r.err, C.c_uint8, n)
        self.out_limit, self.remaining, self.reset_time = (view(getattr(self.r, f), C.c_int64, n) for f in ("limit", "remaining", "reset_time"))
        self._ptrs = {f: getattr(self.b, f) for f in ("burst", "created_at", "is_owner", "behavior", "algorithm")}

    def disable(self, *fields):
        """switch optional request arrays off (NULL pointer: the guber_batch_t default applies)"""
        for f in fields:
            setattr(self.b, f, None)

    def fill(self, hb):
        """copy a HostBatch into the stage (what a batcher does request by request)"""
        n = hb.n
        kb = int(hb.key_off[n])
        self.key_bytes[:kb] = hb.key_bytes[:kb]
        self.key_off[:n + 1] = hb.key_off[:n + 1]
        self.hits[:n] = hb.hits[:n]; self.limit[:n] = hb.limit[:n]; self.duration[:n] = hb.duration[:n]
        for f in ("burst", "created_at", "algorithm", "behavior", "is_owner"):
            src = getattr(hb, f)
            if src is None:
                if f in ("burst",):
                    getattr(self, f)[:n] = 0
                elif f == "created_at":
                    self.created_at[:n] = hb.now_ms
                elif f == "is_owner":
                    self.is_owner[:n] = 1
            else:
                getattr(self, f)[:n] = src[:n]
        self.b.n = n
        self.b.now_ms = hb.now_ms

    def submit(self):
        _check(lib().guber_stage_submit(self.h))

    def wait(self):
        _check(lib().guber_stage_wait(self.h))

    def poll(self):
        """True once the stage's responses are in its result arrays (guber_stage_poll)"""
        L = lib()
        L.guber_stage_poll.argtypes = [C.c_void_p]
        rc = L.guber_stage_poll(self.h)
        if rc < 0:
            _check(rc)
        return rc == 1

    @staticmethod
    def submit_many(stages, aggregates=False):
        """guber_stages_submit: at most one stage per engine; stages of engines that share device and stream share launches"""
        L = lib()
        L.guber_stages_submit.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        arr = (C.c_void_p * len(stages))(*[s.h for s in stages])
        done = C.c_uint32(0)
        _check(L.guber_stages_submit(arr, len(stages), 0 if aggregates else 1, C.byref(done)))
        return done.value

    def route(self, rule, n_engines, timeout_s=10.0):
        """guber_stage_route + guber_stage_route_poll: the device decides every request's engine and rank -> (dest uint32[n], counts)"""
        import time
        L = lib()
        L.guber_stage_route.argtypes = [C.c_void_p, C.POINTER(RouteRule), C.c_uint32]
        L.guber_stage_route_poll.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.guber_stage_dest.argtypes = [C.c_void_p]
        L.guber_stage_dest.restype = C.POINTER(C.c_uint32)
        _check(L.guber_stage_route(self.h, C.byref(rule) if rule is not None else None, n_engines))
        counts = (C.c_uint32 * 16)()
        t0 = time.time()
        while True:
            r = L.guber_stage_route_poll(self.h, counts)
            if r:
                _check(r if r < 0 else 0)
                break
            assert time.time() - t0 < timeout_s, "guber_stage_route did not complete"
        self.engine.synchronize()                              # (dest is complete in stream order; a test reads it from the host)
        dest = np.ctypeslib.as_array(L.guber_stage_dest(self.h), shape=(self.max_n,))[:self.b.n].copy()
        return dest, np.array(counts[:n_engines], np.uint32)

    def submit_routed_as_routed(self, engines, counts):
        """guber_stage_submit_routed with the dest column as guber_stage_route left it"""
        L = lib()
        L.guber_stage_submit_routed.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
        arr = (C.c_void_p * len(engines))(*[e.h for e in engines])
        counts = np.ascontiguousarray(counts, np.uint32)
        _check(L.guber_stage_submit_routed(self.h, arr, len(engines), counts.ctypes.data_as(C.POINTER(C.c_uint32))))

    def submit_routed(self, engines, shard_of, corrupt_dest=None):
        """guber_stage_submit_routed: the stage's requests (already filled, arrival order) belong to several engines — shard_of[i]
        is request i's index in `engines`; ranks inside an engine's share follow the arrival order, as a pool's callers assign them"""
        L = lib()
        L.guber_stage_dest.argtypes = [C.c_void_p]
        L.guber_stage_dest.restype = C.POINTER(C.c_uint32)
        L.guber_stage_submit_routed.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
        n = self.b.n
        shard_of = np.asarray(shard_of, np.uint32)[:n]
        dest = np.ctypeslib.as_array(L.guber_stage_dest(self.h), shape=(self.max_n,))
        counts = np.bincount(shard_of, minlength=len(engines)).astype(np.uint32)
        order = np.argsort(shard_of, kind="stable")
        start = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
        rank = np.empty(n, np.uint32)
        rank[order] = (np.arange(n, dtype=np.int64) - start[shard_of[order]]).astype(np.uint32)
        dest[:n] = (shard_of << 24) | rank
        if corrupt_dest is not None:                            # tests: what a faulty caller might write
            corrupt_dest(dest[:n])
        arr = (C.c_void_p * len(engines))(*[e.h for e in engines])
        _check(L.guber_stage_submit_routed(self.h, arr, len(engines), counts.ctypes.data_as(C.POINTER(C.c_uint32))))
        return counts

    def result(self):
        n = self.b.n
        res = HostResult(n)
        res.status[:n] = self.status[:n]; res.err[:n] = self.err[:n]; res.limit[:n] = self.out_limit[:n]
        res.remaining[:n] = self.remaining[:n]; res.reset_time[:n] = self.reset_time[:n]
        res.c.over_limit_count, res.c.cache_hits, res.c.cache_misses = self.r.over_limit_count, self.r.cache_hits, self.r.cache_misses
        res.c.unexpired_evictions, res.c.cache_size = self.r.unexpired_evictions, self.r.cache_size
        return res

    def close(self):
        if getattr(self, "h", None):
            lib().guber_stage_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass


def set_timezone(offset0_s=0, transitions=()):
    """guber_set_timezone: the daemon's zone for DURATION_IS_GREGORIAN (interval.go uses now.Location()).  offset0_s = UTC offset
    before the first transition; transitions = [(utc_seconds, offset_s_from_then_on), ...] ascending, at most 16.  () and 0 = UTC."""
    class _Tz(C.Structure):
        _fields_ = [("n", C.c_uint32), ("offset0_s", C.c_int32), ("when_s", C.POINTER(C.c_int64)), ("offset_s", C.POINTER(C.c_int32))]
    tr = list(transitions)
    when = (C.c_int64 * max(len(tr), 1))(*[int(w) for w, _ in tr])
    off = (C.c_int32 * max(len(tr), 1))(*[int(o) for _, o in tr])
    tz = _Tz(len(tr), int(offset0_s), when, off)
    L = lib()
    L.guber_set_timezone.argtypes = [C.c_void_p]
    _check(L.guber_set_timezone(C.byref(tz)))


def zone_transitions(name, year_from, year_to):
    """(offset0_s, [(utc_s, offset_s), ...]) of an IANA zone between two years, from the interpreter's zoneinfo — what a binding
    derives from its runtime's zone database and hands to guber_set_timezone"""
    import datetime as dt
    from zoneinfo import ZoneInfo
    z = ZoneInfo(name)
    t = int(dt.datetime(year_from, 1, 1, tzinfo=dt.timezone.utc).timestamp())
    end = int(dt.datetime(year_to + 1, 1, 1, tzinfo=dt.timezone.utc).timestamp())
    off_at = lambda u: int(dt.datetime.fromtimestamp(u, z).utcoffset().total_seconds())
    off0 = cur = off_at(t)
    out = []
    step = 3600
    while t < end:
        if off_at(t + step) != cur:
            lo, hi = t, t + step                                  # bisect the transition to the second
            while hi - lo > 1:
                mid = (lo + hi) // 2
                if off_at(mid) == cur:
                    lo = mid
                else:
                    hi = mid
            cur = off_at(hi)
            out.append((hi, cur))
        t += step
    if len(out) > 16:                                             # (guber_tz_t holds 16: a zone with DST has two per year)
        raise ValueError(f"{name} has {len(out)} transitions between {year_from} and {year_to}: guber_set_timezone takes at most 16 — narrow the window "
                         f"(the daemon needs the years its clock can reach, not history)")
    return off0, out


class V1Instance:
    """The C++ host layer (csrc/worker_pool.h): V1Instance.GetRateLimits over a micro-batching
    GPUWorkerPool.  Thread-safe; requests are dicts with the RateLimitReq field names."""
    ERR_STRIDE = 200

    def __init__(self, cache_size=50_000, device=0, batch_limit=1000, batch_wait_us=500, flags=0, shards=1, devices=None, max_key_bytes=0):
        """devices: list of HIP ordinals = the peers gpu0..gpuN-1 of the replicated consistent hash (the same ordinal may repeat:
        logical devices on one GPU); shards = Config.Workers per device."""
        cfg = GuberConfig(C.sizeof(GuberConfig), device, cache_size, 0, max(batch_limit, 1024), max_key_bytes, None, flags, 0)
        self.h = C.c_void_p()
        L = lib()
        L.guber_pool_create_multi.argtypes = [C.POINTER(GuberConfig), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.guber_pool_shards.argtypes = [C.c_void_p]
        L.guber_pool_shards.restype = C.c_uint32
        L.guber_pool_device_of.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
        L.guber_pool_device_of.restype = C.c_uint32
        L.guber_pool_shard_of.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
        L.guber_pool_shard_of.restype = C.c_uint32
        L.guber_pool_engine_at.argtypes = [C.c_void_p, C.c_uint32]
        L.guber_pool_engine_at.restype = C.c_void_p
        L.guber_pool_metrics.argtypes = [C.c_void_p, C.c_void_p]
        devs = list(devices) if devices else []
        arr = (C.c_int32 * max(len(devs), 1))(*devs) if devs else None
        _check(L.guber_pool_create_multi(C.byref(cfg), arr, len(devs), shards, batch_limit, batch_wait_us, C.byref(self.h)))

    def device_of(self, key):
        kb = key if isinstance(key, bytes) else key.encode()
        return lib().guber_pool_device_of(self.h, kb, len(kb))

    def shard_of(self, key):
        kb = key if isinstance(key, bytes) else key.encode()
        return lib().guber_pool_shard_of(self.h, kb, len(kb))

    def n_shards(self):
        return lib().guber_pool_shards(self.h)

    def shard_size(self, shard):
        return lib().guber_size(lib().guber_pool_engine_at(self.h, shard))

    def metrics(self):
        class M(C.Structure):
            _fields_ = [(f, C.c_uint64) for f in ("batches", "requests", "queue_length", "queue_length_max", "send_duration_us_sum",
                                                  "send_duration_us_max", "batch_size_max", "in_flight", "key_too_long", "flush_on_key_bytes", "rebalances", "keys_moved", "submits", "submit_us_sum", "direct_batches")] + \
                       [("shards", C.c_uint32), ("devices", C.c_uint32)]
        m = M()
        _check(lib().guber_pool_metrics(self.h, C.byref(m)))
        return {f[0]: getattr(m, f[0]) for f in M._fields_}

    def set_clock(self, now_ms):
        lib().guber_pool_set_clock(self.h, now_ms)

    def batches(self):
        return lib().guber_pool_batches(self.h)

    def load(self, items):
        """WorkerPool.Load: items = list of GuberItem (make_item)."""
        arr = (GuberItem * max(len(items), 1))(*items)
        L = lib()
        L.guber_pool_load.argtypes = [C.c_void_p, C.POINTER(GuberItem), C.c_uint32]
        _check(L.guber_pool_load(self.h, arr, len(items)))

    def store(self):
        """WorkerPool.Store: every resident item as a dict (what Loader.Save receives)."""
        out = []
        cb_t = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(GuberItem))
        cb = cb_t(lambda _u, it: out.append(item_dict(it.contents)))
        L = lib()
        L.guber_pool_store.argtypes = [C.c_void_p, cb_t, C.c_void_p]
        _check(L.guber_pool_store(self.h, cb, None))
        return out

    def set_store(self, store):
        """Config.Store (store.go:49-65).  `store` has get(req, key) -> item dict | None, on_change(req, key, item dict),
        remove(req, key), where req is a dict of the request fields the reference hands to the Store (None for remove)."""
        L = lib()
        if store is None:
            L.guber_pool_set_store(self.h, None)
            self._store_cbs = None
            return

        def reqd(q):
            q = q.contents
            key = C.string_at(q.key, q.key_len).decode()
            return key, dict(name=key[:q.name_len], unique_key=key[q.name_len + 1:], hits=q.hits, limit=q.limit, duration=q.duration,
                             burst=q.burst, created_at=q.created_at, algorithm=q.algorithm, behavior=q.behavior)

        def get(_u, q, out):
            key, r = reqd(q)
            d = store.get(r, key)
            if d is None:
                return 0
            o = out.contents
            o.algorithm = d["algorithm"]; o.status = d.get("status", 0); o.limit = d.get("limit", 0); o.duration = d.get("duration", 0)
            o.remaining = d.get("remaining", 0); o.remaining_f = d.get("remaining_f", 0.0); o.stamp = d.get("stamp", 0)
            o.burst = d.get("burst", 0); o.expire_at = d.get("expire_at", 0); o.invalid_at = d.get("invalid_at", 0)
            return 1

        def chg(_u, q, item):
            key, r = reqd(q)
            store.on_change(r, key, item_dict(item.contents, key=key))

        def rem(_u, key, klen):
            store.remove(None, C.string_at(key, klen).decode())
        cbs = abi.GuberStoreCallbacks(abi.STORE_GET_CB(get), abi.STORE_CHG_CB(chg), abi.STORE_REM_CB(rem), None)
        self._store_cbs = cbs
        L.guber_pool_set_store(self.h, C.byref(cbs))

    def global_sync(self):
        """one GlobalSyncWait tick over the pool's devices (guber_pool_global_sync) -> stats dict"""
        from .global_native import SyncStats
        st = SyncStats()
        L = lib()
        L.guber_pool_global_sync.argtypes = [C.c_void_p, C.c_void_p]
        _check(L.guber_pool_global_sync(self.h, C.byref(st)))
        return {f[0]: getattr(st, f[0]) for f in SyncStats._fields_}

    def global_engine_size(self, device):
        L = lib()
        L.guber_pool_global_engine.argtypes = [C.c_void_p, C.c_uint32]
        L.guber_pool_global_engine.restype = C.c_void_p
        return L.guber_size(L.guber_pool_global_engine(self.h, device))

    def add_item(self, item, behavior=None):
        """WorkerPool.AddCacheItem; behavior = the RateLimitReq behaviour the item belongs to (GLOBAL: the device's GLOBAL engine),
        None = where the key already is"""
        L = lib()
        L.guber_pool_add_item.argtypes = [C.c_void_p, C.POINTER(GuberItem)]
        L.guber_pool_add_item_for.argtypes = [C.c_void_p, C.POINTER(GuberItem), C.c_uint32]
        _check(L.guber_pool_add_item(self.h, C.byref(item)) if behavior is None else L.guber_pool_add_item_for(self.h, C.byref(item), behavior))

    def get_item(self, key):
        """WorkerPool.GetCacheItem -> item dict or None"""
        L = lib()
        L.guber_pool_get_item.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.POINTER(GuberItem), C.POINTER(C.c_int)]
        kb = key if isinstance(key, bytes) else key.encode()
        out, found = GuberItem(), C.c_int(0)
        _check(L.guber_pool_get_item(self.h, kb, len(kb), C.byref(out), C.byref(found)))
        return abi.item_dict(out, kb) if found.value else None

    def size(self):
        L = lib()
        L.guber_pool_size.argtypes = [C.c_void_p]
        L.guber_pool_size.restype = C.c_int64
        return L.guber_pool_size(self.h)

    def GetRateLimits(self, reqs, is_owner=None):
        """-> list of dicts {status, limit, remaining, reset_time, error}; raises GuberError for the
        RPC-level OutOfRange error (more than 1000 requests).  is_owner: RateLimitReqState.IsOwner per request (None = all)."""
        n = len(reqs)
        def strs(field):
            bs = [r.get(field, "").encode() for r in reqs]
            off = np.zeros(n + 1, np.uint32)
            if n:
                off[1:] = np.cumsum([len(b) for b in bs])
            return np.frombuffer(b"".join(bs) + b"\0", np.uint8).copy(), off
        nb, no = strs("name")
        kb, ko = strs("unique_key")
        col = lambda f, dt: np.array([r.get(f, 0) for r in reqs], dtype=dt) if n else np.zeros(1, dt)
        hits, limit, duration, burst, created = (col(f, np.int64) for f in ("hits", "limit", "duration", "burst", "created_at"))
        algo, beh = col("algorithm", np.int32), col("behavior", np.uint32)
        res = HostResult(n)
        txt = C.create_string_buffer(max(n, 1) * self.ERR_STRIDE)
        if is_owner is not None:
            own = np.array([1 if x else 0 for x in is_owner], np.uint8) if n else np.zeros(1, np.uint8)
            rc = lib().guber_pool_get_rate_limits_owner(self.h, n, nb.ctypes.data, no.ctypes.data, kb.ctypes.data, ko.ctypes.data,
                                                        hits.ctypes.data, limit.ctypes.data, duration.ctypes.data, burst.ctypes.data,
                                                        created.ctypes.data, algo.ctypes.data, beh.ctypes.data, own.ctypes.data, C.byref(res.c), txt,
                                                        self.ERR_STRIDE)
        else:
            rc = lib().guber_pool_get_rate_limits(self.h, n, nb.ctypes.data, no.ctypes.data, kb.ctypes.data, ko.ctypes.data,
                                                  hits.ctypes.data, limit.ctypes.data, duration.ctypes.data, burst.ctypes.data,
                                                  created.ctypes.data, algo.ctypes.data, beh.ctypes.data, C.byref(res.c), txt,
                                                  self.ERR_STRIDE)
        if rc != 0:
            raise GuberError(rc, txt.raw[:self.ERR_STRIDE].split(b"\0")[0].decode())
        out = []
        for i in range(n):
            err = txt.raw[i * self.ERR_STRIDE:(i + 1) * self.ERR_STRIDE].split(b"\0")[0].decode()
            out.append(dict(status=int(res.status[i]), limit=int(res.limit[i]), remaining=int(res.remaining[i]),
                            reset_time=int(res.reset_time[i]), error=err))
        return out

    def close(self):
        if getattr(self, "h", None):
            lib().guber_pool_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
