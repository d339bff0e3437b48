"""Data containers of the C ABI's GLOBAL rows (guber_global_rows_t, guber_item_t) as numpy arrays: what Engine.global_take returns
and Engine.add_items_struct accepts.  The exchange itself is native (csrc/guber_global_sync.h: guber_comm_*, guber_global_sync)."""
import ctypes as C

import numpy as np

from . import abi

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


