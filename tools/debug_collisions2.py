import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, xxhash, collections
import gubernator_amd as ga, support, streams
from support import HostBatch, Oracle
# two keys with the same weak tag, a few requests each, careful-only engine
tag = lambda k: xxhash.xxh64(k.encode(), seed=0).intdigest() & 0x1f80
by = collections.defaultdict(list)
for i in range(200):
    by[tag(f"coll_{i}")].append(f"coll_{i}")
pair = next(v for v in by.values() if len(v) >= 2)[:2]
print("pair", pair, hex(tag(pair[0])))
e = ga.Engine(cache_size=4096, max_batch=4096, flags=5)
o = Oracle(cache_size=1 << 16)
keys = [pair[0], pair[1], pair[0], pair[1], pair[1], pair[0], pair[1]]
b = HostBatch(keys, 1, 7, 2000, streams.NOW0)
L = ga.lib()
L.gbdbg_read_work.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32]
got, want = e.eval(b), o.eval(b)
print("got ", got.rows())
print("want", want.rows())
n = 8
for w, name in enumerate(["did", "slot", "rflags", "lrank", "seg_slot", "seg_flags"]):
    a = np.zeros(n, np.uint32)
    L.gbdbg_read_work(e.h, w, a.ctypes.data, n)
    print(name, a.tolist())
print(e.stats())
print(sorted(d["key"] for d in e.each()))
