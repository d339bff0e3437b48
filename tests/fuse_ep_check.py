#!/usr/bin/env python
"""The fused launch k_evalpart_multi — one group's k_eval3 + the same tables' next k_part (gubernator_amd/csrc/guber_kernels_part.h,
launch_group in guber_engine.hip; the engine's default since round 5's first GPU call measured it: +3.5 % on the headline) — on the GPU
against the oracle.  Run by tests/test_gpu_parity.py::test_k_eval3_and_the_next_k_part_share_a_launch in a process of its own, or by hand:
  python tests/fuse_ep_check.py
Four tables on one stream (then six: two groups per round), 30 rounds of one batch each through ONE guber_eval_batches_routed_dev call: adversarial Zipf batches with
hot keys, both algorithms, the clock stepping so that buckets expire and renew; in between a batch too small for the owner-partitioned
pipeline (the held-back k_eval3 must go first), a round in which one table has no batch (another group: flush), uniform keys that make
the owner count move (one batch later than without the fusion).  Every batch must equal its table's oracle; k_evalpart_multi must have
been launched (profile_read)."""
import ctypes as C
import os
import sys

os.environ.setdefault("GUBER_FUSE_EP", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import numpy as np
import torch

import gubernator_amd as ga
import streams
import support


def main(NE=4):
    dev = torch.device("cuda", 0)
    K, B, rounds = 6000, 8192, 30
    tab = streams.key_table(K * NE)
    stream = torch.cuda.Stream(device=dev)
    engs = [ga.Engine(cache_size=1 << 18, max_batch=65536, stream=stream.cuda_stream) for _ in range(NE)]
    orcs = [support.Oracle(cache_size=1 << 18) for _ in range(NE)]
    zs = [streams.ZipfSampler(K, seed=300 + j) for j in range(NE)]
    rng = np.random.default_rng(5)
    for e in engs:
        e.profile(True)                                    # (a group's launches are timed by the group's first engine)
    which, hbs, keep, cb, cr = [], [], [], [], []
    for r in range(rounds):
        for j in range(NE):
            if r == 11 and j == 2:
                continue                                   # a round without table 2: a different group -> the held-back k_eval3 goes first
            n = 300 if (r == 7 and j == 1) else [B, B, 5000, B, 2 * B][(r + j) % 5]      # 300: not the owner-partitioned pipeline -> flush
            ids = rng.integers(0, K, n) if r in (15, 16, 17) else zs[j].draw(n)          # uniform keys: rounds split, the owner count moves
            hb = streams.bench_batch(tab, j * K + ids, streams.NOW0 + r * 900, algorithm=(r + j) % 2, limit=30, duration=4000)
            t = [torch.from_numpy(hb.key_bytes).to(dev), torch.from_numpy(hb.key_off.view(np.int32)).to(dev),
                 torch.from_numpy(hb.hits).to(dev), torch.from_numpy(hb.limit).to(dev), torch.from_numpy(hb.duration).to(dev),
                 torch.from_numpy(hb.algorithm).to(dev), torch.from_numpy(hb.behavior.view(np.int32)).to(dev)]
            p = [x.data_ptr() for x in t]
            res = dict(status=torch.empty(n, dtype=torch.uint8, device=dev), err=torch.empty(n, dtype=torch.uint8, device=dev),
                       limit=torch.empty(n, dtype=torch.int64, device=dev), remaining=torch.empty(n, dtype=torch.int64, device=dev),
                       reset_time=torch.empty(n, dtype=torch.int64, device=dev))
            keep.append((t, res)); which.append(j); hbs.append(hb)
            cb.append(ga.GuberBatch(n, 0, p[0], p[1], p[2], p[3], p[4], None, None, p[5], p[6], None, None, None, hb.now_ms))
            cr.append(ga.GuberResult(res["status"].data_ptr(), res["limit"].data_ptr(), res["remaining"].data_ptr(),
                                     res["reset_time"].data_ptr(), res["err"].data_ptr(), 0, 0, 0, 0, 0))
    torch.cuda.synchronize(dev)
    N = len(which)
    ga.Engine.eval_routed_dev(engs, (C.c_uint32 * N)(*which), (ga.GuberBatch * N)(*cb), (ga.GuberResult * N)(*cr), N)
    for e in engs:
        e.synchronize()
    bad = 0
    for s in range(N):
        want = orcs[which[s]].eval(hbs[s])
        got = ga.HostResult(hbs[s].n)
        for name in ("status", "limit", "remaining", "reset_time", "err"):
            getattr(got, name)[:] = keep[s][1][name].cpu().numpy()
        try:
            support.assert_results_equal(got, want, f"batch {s} of table {which[s]}")
        except AssertionError as ex:
            bad += 1
            print("MISMATCH", str(ex)[:400])
    prof = {}
    for e in engs:
        for k, v in e.profile_read().items():
            prof[k] = (prof.get(k, (0, 0.0))[0] + v[0], prof.get(k, (0, 0.0))[1] + v[1])
    print("launches:", {k: v[0] for k, v in prof.items() if v[0]})
    for j, (e, o) in enumerate(zip(engs, orcs)):
        if e.size() != o.size():
            bad += 1
            print("SIZE", j, e.size(), o.size())
        e.close()
    fused = prof.get("k_evalpart_multi", (0, 0))[0]
    if os.environ.get("GUBER_FUSE_EP", "1") == "1" and fused < (rounds // 2) * ((NE + 3) // 4):
        bad += 1
        print("k_evalpart_multi was launched", fused, "times only")
    print(f"{NE} tables on one stream:", "ok" if not bad else f"FAILED ({bad})")
    return bad


if __name__ == "__main__":
    # four tables = one group per round; six = two groups per round on one stream (two k_eval3 held back at a time: PendSet)
    bad = main(4) + main(6)
    print("FUSE_EP CHECK", "OK" if not bad else f"FAILED ({bad})")
    sys.exit(1 if bad else 0)
