import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, xxhash, collections
import gubernator_amd as ga, support, streams
from support import HostBatch, Oracle
nkeys = int(sys.argv[1]) if len(sys.argv) > 1 else 400
nreq = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
o, e = Oracle(cache_size=1 << 16), ga.Engine(cache_size=4096, max_batch=4096, flags=int(sys.argv[3]) if len(sys.argv) > 3 else 1)
rng = np.random.default_rng(5)
now = streams.NOW0
ids = rng.integers(0, nkeys, nreq)
keys = [f"coll_{int(i)}" for i in ids]
b = HostBatch(keys, 1, 7, 2_000, now, algorithm=(ids % 2).astype(np.uint8))
got, want = e.eval(b), o.eval(b)
st = e.stats()
print("stats", st)
bad = np.nonzero((got.status[:nreq] != want.status[:nreq]) | (got.remaining[:nreq] != want.remaining[:nreq]) | (got.err[:nreq] != want.err[:nreq]))[0]
print("bad", len(bad))
tag = lambda k: xxhash.xxh64(k.encode(), seed=0).intdigest() & 0x1f80
chain = collections.Counter(tag(f"coll_{i}") for i in range(nkeys))
badkeys = collections.Counter(keys[i] for i in bad)
print("distinct bad keys", len(badkeys), "of", len(set(keys)))
for k, c in list(badkeys.items())[:12]:
    idx = [i for i in range(nreq) if keys[i] == k]
    print(k, "tag", hex(tag(k)), "chain", chain[tag(k)], "reqs", len(idx), "bad", c,
          "got", [(int(got.status[i]), int(got.remaining[i]), int(got.err[i])) for i in idx[:10]],
          "want", [(int(want.status[i]), int(want.remaining[i])) for i in idx[:10]])
# do sizes agree, and which keys are resident
print("sizes", e.size(), o.size())
items = {d["key"]: d for d in e.each()}
oitems = {d["key"]: d for d in o.each()}
diff = [k for k in oitems if k not in items or items[k]["remaining"] != oitems[k]["remaining"] or items[k]["remaining_f"] != oitems[k]["remaining_f"]]
print("state diffs", len(diff), diff[:10])
for k in diff[:5]:
    print(k, items.get(k), oitems[k])
