import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import gubernator_amd as ga, support, streams
from support import HostBatch
L = ga.lib()
L.gbdbg_read_work.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32]
def dump(e, n=4):
    out = {}
    for w, name in enumerate(["did", "slot", "rflags", "lrank", "seg_slot", "seg_flags"]):
        a = np.zeros(n, np.uint32); L.gbdbg_read_work(e.h, w, a.ctypes.data, n); out[name] = a.tolist()
    return out
for flags in (4, 0, 5, 1):
    print("=== flags", flags)
    e = ga.Engine(cache_size=4096, max_batch=4096, flags=flags)
    for step, keys in enumerate([["coll_0"] * 3, ["coll_38"] * 3, ["coll_0", "coll_38", "coll_0"]]):
        r = e.eval(HostBatch(keys, 1, 7, 2000, streams.NOW0))
        print(step, keys[0], r.rows(), dump(e), "size", e.size(), "each", sorted(d["key"] for d in e.each()), "batches", e.stats()["batches"])
