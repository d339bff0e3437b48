import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gubernator_amd as ga, streams, support
from support import HostBatch, Oracle
for flags in (32, 0):
    rng = np.random.default_rng(41)
    o, e = Oracle(cache_size=1 << 16), ga.Engine(cache_size=4096, max_batch=4096, flags=flags)
    now = streams.NOW0
    bad = 0
    for step in range(400):
        n = int(rng.choice([1, 1, 2, 3, 17, 64, 255, 256]))
        ids = rng.integers(0, 30, n)
        keys = [b"" if rng.random() < 0.01 else b"small_%d" % int(i) for i in ids]
        uniform = rng.random() < 0.7
        hits = np.full(n, int(rng.choice([0, 1, 2, 5]))) if uniform else rng.choice([0, 1, 2, 5], n)
        algo = (ids % 2).astype(np.uint8) if rng.random() < 0.9 else rng.choice([0, 1, 7], n).astype(np.uint8)
        b = HostBatch(keys, hits, 20, int(rng.choice([50, 5000])), now, algorithm=algo, behavior=int(rng.choice([0, 0, 32])))
        got, want = e.eval(b), o.eval(b)
        g, w = got.rows(), want.rows()
        diff = [i for i in range(n) if g[i] != w[i]]
        if diff and bad < 3:
            bad += 1
            print(f"flags {flags} step {step} n {n} uniform {uniform}: {len(diff)} diffs at {diff[:10]}")
            for i in diff[:4]:
                same = [j for j in range(n) if keys[j] == keys[i]]
                print("   idx", i, "key", keys[i], "got", g[i], "want", w[i], "same-key idx", same[:12], "hits", [int(hits[j]) for j in same[:12]], "algo", [int(algo[j]) for j in same[:12]])
            empt = [j for j in range(n) if keys[j] == b""]
            print("   empty keys at", empt)
        now += int(rng.choice([0, 1, 40, 6000]))
    print("flags", flags, "bad batches", bad)
