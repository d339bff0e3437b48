#!/usr/bin/env python3
"""bench.py — rate-limit decisions/sec of the MI355X engine on BASELINE.json's workload.

One "step" = one GetRateLimits batch of 65536 checks evaluated by the HIP path (guber_eval_batch_dev: k_front ->
k_eval2) with every input array already resident in HBM.  Workload (BASELINE.json configs[1]):
10M resident keys per GPU, ONE Zipf(1.1) request stream over those keys (stream seed 1234, permutation seed 99),
TOKEN_BUCKET, hits 1, limit 100, duration 60 s, now_ms advancing 1 ms per batch.

Inside a GPU the resident keys are split into S logical shards (default 12; the reference shards its key space the same
way over Config.Workers goroutines, workers.go:19-25): S engines with their own HBM tables.  The request
stream is routed request by request to the shard that holds the key (load-aware hash slots, gubernator_amd/shard.py
SlotMap: 256 slots by consistent hash / k_route, slots and the few hottest keys placed on the shards by what a sample of
earlier traffic carried; --router ring = a plain consistent hash over the shards) and every shard flushes a
batch when 65536 requests are waiting — the policy of the reference's batcher (peer_client.go:284-337) — so batches are
exactly 65536 requests, hot shards flush more often, and per-key request order is the stream's order.

Timing: the batches of the timed region are enqueued by ONE dispatcher (guber_eval_batches_routed_dev: round by round the
next batch of every shard; shards that share a stream — 12 shards over 3 streams by default — share their two launches,
k_front_multi / k_eval2_multi), or with --dispatch threads by S pre-started batcher threads, one per shard and stream
(released by a barrier; no thread is created and nothing is allocated inside the timed region).  The region is repeated
until it lasts at least --min-ms, whatever --steps says.  Extras in the same JSON line: `leaky` (configs[2],
parity-gated), `shards_1` (one table, the literal single-stream configuration), `uniform` (no duplicate keys),
`end_to_end` (host pointers in, host results out, PCIe included), `pool` (caller threads -> V1Instance::GetRateLimits ->
the C++ GPUWorkerPool).

N > 1 (launched by torch.distributed.run, one rank per GPU): the key space is N x 10M keys sharded by the reference's
replicated consistent hash (replicated_hash.go; 512 vnodes, fnv1, peers gpu0..gpuN-1), every rank evaluates the requests
for the keys it owns — no data-path collective (weak scaling).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0           # MI355X spec (MI355X_MICROARCH.md)
BYTES_PER_DECISION = {"token": 149, "leaky": 173}   # SURVEY.md section 8d, 16-byte keys
# split of the algorithmic bytes over the kernels that touch request / table / response data (DESIGN.md
# "Algorithmic bytes"): k_front reads key_off 4 + key 16 + table 56 (token) / 64 (leaky); k_eval2 reads
# the request fields 32 / 40, writes table 16 / 24 and the response 25.
KERNEL_BYTES = {"token": {"k_front": 76, "k_eval2": 73, "k_front_multi": 76, "k_eval2_multi": 73, "k_resolve": 28, "k_eval": 121},
                "leaky": {"k_front": 84, "k_eval2": 89, "k_front_multi": 84, "k_eval2_multi": 89, "k_resolve": 28, "k_eval": 145}}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--keys", type=int, default=10_000_000, help="resident keys per GPU")
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--algo", choices=["token", "leaky"], default="token")
    ap.add_argument("--dist", choices=["zipf", "uniform"], default="zipf")
    ap.add_argument("--min-ms", type=float, default=250.0, help="minimum duration of the timed region: the timed steps are repeated until it is reached")
    ap.add_argument("--extras", default="leaky,shards_1,uniform,end_to_end,pool",
                    help="comma list of extra configurations measured after the headline one (N = 1 only); '' = none")
    ap.add_argument("--dispatch", choices=["threads", "one"], default="one",
                    help="who enqueues the shards' batches: one pre-started thread per shard, or ONE dispatcher for all shards in flush order "
                         "(guber_eval_batches_routed_dev)")
    ap.add_argument("--router", choices=["slots", "ring"], default=None,
                    help="placement of a GPU's keys on its logical shards: load-aware hash slots (shard.SlotMap; default with --dispatch one) or "
                         "a plain consistent hash (default with --dispatch threads)")
    ap.add_argument("--streams", type=int, default=3, help="with --dispatch one: streams the shards are spread over (shards of one stream share launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="CPU time budget per thread count of the baseline")
    ap.add_argument("--cpu-threads", default="1,32,all", help="worker shards / threads of the CPU baseline (comma list, 'all' = every host core)")
    ap.add_argument("--profile-steps", type=int, default=32)
    ap.add_argument("--shards", type=int, default=12, metavar="S",
                    help="logical key-space shards per GPU (the reference's Config.Workers sharding, workers.go:19-25): S "
                         "engines with their own tables, the stream routed to them key by key")
    ap.add_argument("--global-host", action="store_true", help="with --global-sync: use the host-staged exchange (global_sync.py)")
    ap.add_argument("--global-sync", type=int, default=0, metavar="K",
                    help="BASELINE config 5: every request carries GLOBAL, every rank serves ALL keys from its replica, "
                         "and every K steps the ranks exchange pending hits / broadcast owner state (0 = off)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL)")
    ap.add_argument("--one-device", action="store_true",
                    help="debug: all ranks share GPU 0 (single-GPU box; use with --backend gloo)")
    return ap.parse_args()


class Ctx:
    """process-wide state: torch device, distributed world, key table"""


def percentile(sorted_vals, p):
    return sorted_vals[min(len(sorted_vals) - 1, int(len(sorted_vals) * p))]


class BatcherThreads:
    """S pre-started threads, one per logical shard (the batcher goroutine of a shard).  run(jobs) releases them through a
    barrier, each executes its job (enqueue its batches), and a second barrier collects them: nothing is created inside
    the timed region."""

    def __init__(self, n):
        self.n = n
        self.start = threading.Barrier(n + 1)
        self.end = threading.Barrier(n + 1)
        self.jobs = [None] * n
        self.err = [None] * n
        self.stop = False
        self.threads = [threading.Thread(target=self._loop, args=(j,), daemon=True) for j in range(n)]
        for t in self.threads:
            t.start()

    def _loop(self, j):
        while True:
            self.start.wait()
            if self.stop:
                return
            try:
                if self.jobs[j] is not None:
                    self.jobs[j]()
            except Exception as ex:   # noqa: BLE001
                self.err[j] = ex
            self.end.wait()

    def run(self, jobs):
        self.jobs = list(jobs)
        self.start.wait()
        self.end.wait()
        for ex in self.err:
            if ex is not None:
                raise ex

    def close(self):
        self.stop = True
        self.start.wait()
        for t in self.threads:
            t.join()


class Rig:
    """One measured configuration: S engines (tables + streams) over this rank's keys, a routed request sequence resident
    in HBM, and the machinery to run it."""

    def __init__(self, ctx, algo, dist_kind, S, flags=0, max_key_bytes=0):
        import torch
        import gubernator_amd as ga
        import streams
        from gubernator_amd import shard
        self.ctx, self.algo, self.dist_kind, self.S = ctx, algo, dist_kind, S
        self.dispatch = getattr(ctx, "dispatch", "threads")
        self.algo_id = 0 if algo == "token" else 1
        self.torch, self.ga, self.streams = torch, ga, streams
        dev, K, B = ctx.dev, ctx.K, ctx.B
        self.sstreams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        if self.dispatch == "one":             # one dispatcher: shards that share a stream share launches
            ns = max(1, min(S, int(getattr(ctx, "streams", 1))))
            self.sstreams = [self.sstreams[j * ns // S] for j in range(S)]
        nk = len(ctx.my_ids)
        # logical shards inside this GPU: the rank's keys are placed on them first (k_route on a scratch engine), then every
        # shard gets a table for the keys it really holds (+ 25 %)
        router_engine = ga.Engine(cache_size=1024, device=ctx.local_rank, max_batch=1024) if S > 1 else None
        self.engines = [router_engine]
        self.placement = None
        if S > 1 and getattr(ctx, "router", "slots") == "slots":
            # load-aware placement (gubernator_amd/shard.py SlotMap): keys -> 256 hash slots (k_route), slots and the few keys
            # that alone outweigh a slot -> shards by what an earlier sample of the traffic carried
            sm = shard.SlotMap(S)
            slot_of = np.concatenate([ctx.route_on_device(self.engines[0], sm.ring, *streams.keys_for_ids(ctx.table, ctx.my_ids[lo:lo + 4_000_000]))
                                      for lo in range(0, nk, 4_000_000)]).astype(np.int64)
            observed = (streams.ZipfSampler(nk, s=1.1, seed=990_001 + ctx.rank, perm_seed=99).draw(1 << 21) if dist_kind == "zipf"
                        else np.zeros(0, np.int64))
            self.sown = sm.place(slot_of, observed).astype(np.uint8)
            self.placement = {"router": "256 hash slots placed on the shards by observed load (2 M earlier requests), hot keys individually",
                              "keys_placed_individually": int(len(sm.hot_ids)), "expected_share_max": round(float(sm.load.max()), 4),
                              "expected_share_min": round(float(sm.load.min()), 4)}
            sm.close()
        elif S > 1:
            sring = ga.Ring([f"gpu{ctx.rank}-shard{j}" for j in range(S)], 512, "fnv1")
            self.sown = np.concatenate([ctx.route_on_device(self.engines[0], sring, *streams.keys_for_ids(ctx.table, ctx.my_ids[lo:lo + 4_000_000]))
                                        for lo in range(0, nk, 4_000_000)]).astype(np.uint8)
            sring.close()
            self.placement = {"router": "replicated consistent hash over the shards (512 vnodes, fnv1)"}
        else:
            self.sown = np.zeros(nk, np.uint8)
        self.local_of_shard = [np.nonzero(self.sown == j)[0] for j in range(S)]
        if router_engine is not None:
            router_engine.close()
        self.engines = [ga.Engine(cache_size=len(self.local_of_shard[j]) + len(self.local_of_shard[j]) // 4 + 1024, device=ctx.local_rank, max_batch=B,
                                  stream=self.sstreams[j].cuda_stream, max_key_bytes=max_key_bytes, flags=flags) for j in range(S)]
        # arrays every batch of this rig shares (fixed-width keys, constant request fields)
        L = ctx.table.shape[1]
        self.t_off = torch.from_numpy((np.arange(B + 1, dtype=np.int64) * L).astype(np.int32)).to(dev)
        self.t_hits1 = torch.full((B,), 1, dtype=torch.int64, device=dev)
        self.t_hits0 = torch.zeros((B,), dtype=torch.int64, device=dev)
        self.t_limit = torch.full((B,), 100, dtype=torch.int64, device=dev)
        self.t_dur = torch.full((B,), 60_000, dtype=torch.int64, device=dev)
        self.t_algo = torch.full((B,), self.algo_id, dtype=torch.uint8, device=dev)
        self.t_beh = torch.full((B,), 2 if (flags & ga.FLAG_GLOBAL) else 0, dtype=torch.int32, device=dev)
        self.scratch = [self.DevResult(self, B) for _ in range(S)]
        self.workers = BatcherThreads(S) if S > 1 else None
        self.keep = []          # tensors referenced by C structs

    class DevResult:
        def __init__(self, rig, n):
            torch, ga, dev = rig.torch, rig.ga, rig.ctx.dev
            self.status = torch.empty(n, dtype=torch.uint8, device=dev)
            self.err = torch.empty(n, dtype=torch.uint8, device=dev)
            self.limit = torch.empty(n, dtype=torch.int64, device=dev)
            self.remaining = torch.empty(n, dtype=torch.int64, device=dev)
            self.reset_time = torch.empty(n, dtype=torch.int64, device=dev)
            self.c = ga.GuberResult(self.status.data_ptr(), self.limit.data_ptr(), self.remaining.data_ptr(),
                                    self.reset_time.data_ptr(), self.err.data_ptr(), 0, 0, 0, 0, 0)
            self.ga = ga

        def host(self):
            h = self.ga.HostResult(len(self.status))
            for name in ("status", "limit", "remaining", "reset_time", "err"):
                getattr(h, name)[:] = getattr(self, name).cpu().numpy()
            return h

    def dev_batch(self, ids, now_ms, hits=1, owner_ptr=None):
        """GuberBatch over global key ids `ids` (device pointers); len(ids) <= B"""
        torch, ga = self.torch, self.ga
        kb, _ = self.streams.keys_for_ids(self.ctx.table, ids)
        t_keys = torch.from_numpy(kb).to(self.ctx.dev)
        self.keep.append(t_keys)
        return ga.GuberBatch(len(ids), 0, t_keys.data_ptr(), self.t_off.data_ptr(),
                             (self.t_hits1 if hits else self.t_hits0).data_ptr(), self.t_limit.data_ptr(), self.t_dur.data_ptr(),
                             None, None, self.t_algo.data_ptr(), self.t_beh.data_ptr(), owner_ptr, None, None, int(now_ms))

    def populate(self, now0):
        """residency: every owned key gets a bucket before anything is timed (hits 0 = create, consume nothing)"""
        B = self.ctx.B
        for j in range(self.S):
            ids = self.ctx.my_ids[self.local_of_shard[j]]
            for lo in range(0, len(ids), B):
                b = self.dev_batch(ids[lo:lo + B], now0, hits=0)
                self.engines[j].eval_dev(b, self.scratch[j].c)
                self.engines[j].synchronize()
                self.keep.pop()
        return sum(e_.size() for e_ in self.engines)

    def build_sequence(self, total, now0, seed):
        """Draw ONE request stream over this rank's keys, route it to the shards, flush a shard's batch whenever B requests
        are waiting (batches in flush order).  -> list of (shard, global ids of the batch, now_ms)."""
        B, S = self.ctx.B, self.S
        nk = len(self.ctx.my_ids)
        if self.dist_kind == "zipf":
            smp = self.streams.ZipfSampler(nk, s=1.1, seed=seed, perm_seed=99)
            draw = smp.draw
        else:
            rg = np.random.default_rng(seed)
            draw = (lambda n: rg.permutation(nk)[:n]) if B * S <= nk else (lambda n: rg.integers(0, nk, n))
        pend_ids = [np.zeros(0, np.int64) for _ in range(S)]
        pend_pos = [np.zeros(0, np.int64) for _ in range(S)]
        out, offset = [], 0
        while len(out) < total + S:        # a few more than needed, then cut in flush order
            li = draw(B * S)
            sh = self.sown[li]
            pos = offset + np.arange(len(li), dtype=np.int64)
            offset += len(li)
            for j in range(S):
                m = sh == j if S > 1 else slice(None)
                pend_ids[j] = np.concatenate([pend_ids[j], li[m]])
                pend_pos[j] = np.concatenate([pend_pos[j], pos[m]])
                while len(pend_ids[j]) >= B:
                    out.append((int(pend_pos[j][B - 1]), j, pend_ids[j][:B]))
                    pend_ids[j], pend_pos[j] = pend_ids[j][B:], pend_pos[j][B:]
        out.sort(key=lambda x: x[0])
        return [(j, self.ctx.my_ids[li], now0 + 1 + s) for s, (_, j, li) in enumerate(out[:total])]

    def load_sequence(self, seq, keep_first):
        """device batches + result targets for a sequence; the first `keep_first` batches keep their results"""
        B = self.ctx.B
        self.seq = seq
        self.batches = [self.dev_batch(ids, now) for (_, ids, now) in seq]
        self.kept = [self.DevResult(self, B) for _ in range(min(keep_first, len(seq)))]

    def _arrays(self, lo, hi, keep):
        """per shard: ctypes arrays (GuberBatch[], GuberResult[], count) of the sequence's batches lo..hi in order"""
        ga = self.ga
        per = []
        for j in range(self.S):
            idx = [s for s in range(lo, hi) if self.seq[s][0] == j]
            ba = (ga.GuberBatch * max(len(idx), 1))(*[self.batches[s] for s in idx])
            ra = (ga.GuberResult * max(len(idx), 1))(*[(self.kept[s].c if (keep and s < len(self.kept)) else self.scratch[j].c) for s in idx])
            per.append((ba, ra, len(idx)))
        return per

    def _routed(self, lo, hi, keep):
        """the sequence's batches lo..hi in flush order for ONE dispatcher: (which[], GuberBatch[], GuberResult[], count)"""
        import ctypes as C
        ga = self.ga
        idx = list(range(lo, hi))
        wa = (C.c_uint32 * max(len(idx), 1))(*[self.seq[s][0] for s in idx])
        ba = (ga.GuberBatch * max(len(idx), 1))(*[self.batches[s] for s in idx])
        ra = (ga.GuberResult * max(len(idx), 1))(*[(self.kept[s].c if (keep and s < len(self.kept)) else self.scratch[self.seq[s][0]].c) for s in idx])
        return wa, ba, ra, len(idx)

    def run(self, lo, hi, repeats=1, keep=False, timed=False):
        """enqueue batches lo..hi of the sequence `repeats` times.  timed: returns (wall seconds, max per-stream event ms)"""
        torch = self.torch
        per = self._arrays(lo, hi, keep)
        if self.dispatch == "one" and self.S > 1:
            wa, ba1, ra1, cnt1 = self._routed(lo, hi, keep)

        def job(j):
            ba, ra, cnt = per[j]
            eng = self.engines[j]

            def f():
                for _ in range(repeats):
                    if cnt:
                        eng.eval_many_dev(ba, ra, cnt)
            return f
        jobs = [job(j) for j in range(self.S)]
        ev0 = ev1 = None
        if timed:
            ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(self.S)]
            ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(self.S)]
            torch.cuda.synchronize(self.ctx.dev)
            self.ctx.barrier()
            for j in range(self.S):
                ev0[j].record(self.sstreams[j])
        t0 = time.perf_counter()
        if self.dispatch == "one" and self.S > 1:
            for _ in range(repeats):
                self.ga.Engine.eval_routed_dev(self.engines, wa, ba1, ra1, cnt1)
        elif self.workers is not None:
            self.workers.run(jobs)
        else:
            jobs[0]()
        if timed:
            for j in range(self.S):
                ev1[j].record(self.sstreams[j])
        torch.cuda.synchronize(self.ctx.dev)
        t1 = time.perf_counter()
        if timed:
            self.ctx.barrier()
            self.last_stream_ms = [ev0[j].elapsed_time(ev1[j]) for j in range(self.S)]
            self.last_stream_batches = [per[j][2] * repeats for j in range(self.S)]
            return t1 - t0, max(self.last_stream_ms)
        return t1 - t0, None

    def measure(self, steps, warmup, min_ms, now0, seed, keep_first=8):
        """warm up, calibrate the repeat count, time.  -> dict"""
        ctx = self.ctx
        seq = self.build_sequence(warmup + steps, now0, seed)
        self.load_sequence(seq, keep_first)
        if warmup:
            self.run(0, warmup, keep=True)
        cal, _ = self.run(warmup, warmup + steps, keep=True)          # first execution of the timed steps: calibration (untimed)
        cal = ctx.max_over_ranks(cal)
        repeats = max(1, int(math.ceil(min_ms * 1e-3 / max(cal, 1e-6))))
        for _ in range(3):                     # the calibration pass runs cold: repeat until the timed region really lasts min_ms
            wall, ev_ms = self.run(warmup, warmup + steps, repeats=repeats, timed=True)
            wall = ctx.max_over_ranks(wall)
            if wall * 1e3 >= 0.95 * min_ms:
                break
            repeats = int(math.ceil(repeats * min_ms * 1e-3 / wall * 1.1))
        n_batches = steps * repeats
        return {"value": n_batches * ctx.B * ctx.world / wall, "ms_per_step": wall / n_batches * 1e3, "repeats": repeats,
                "timed_ms": wall * 1e3, "ms_per_step_events": ev_ms / n_batches, "steps": steps,
                "shard_streams": [{"batches": b, "stream_ms": round(m, 3), "us_per_batch": round(m * 1e3 / max(b, 1), 2)}
                                  for b, m in zip(self.last_stream_batches, self.last_stream_ms)]}

    def kernel_profile(self, n_steps, lo, hi):
        """per-kernel durations with one batch in flight on shard 0 (HIP events around every launch)"""
        eng = self.engines[0]
        idx = [s for s in range(lo, hi) if self.seq[s][0] == 0] or [lo]
        eng.profile(True)
        eng.profile_read()
        for k in range(n_steps):
            eng.eval_dev(self.batches[idx[k % len(idx)]], self.scratch[0].c)
        prof = eng.profile_read()
        eng.profile(False)
        return {k: (ms / n if n else 0.0) for k, (n, ms) in prof.items()}

    def kernel_profile_routed(self, lo, hi):
        """one dispatcher: the timed steps once more with HIP events around every (fused) launch, all shards overlapping as in
        the timed region.  -> ({kernel: avg ms per launch}, {kernel: avg requests per launch})"""
        for e in self.engines:
            e.profile(True)
            e.profile_read()
        self.run(lo, hi)
        ms, n, units = {}, {}, {}
        for e in self.engines:
            prof = e.profile_read()
            e.profile(False)
            for k, (cnt, tot) in prof.items():
                n[k] = n.get(k, 0) + cnt
                ms[k] = ms.get(k, 0.0) + tot
                units[k] = units.get(k, 0) + e.last_profile_units.get(k, 0)
        return ({k: ms[k] / n[k] for k in n if n[k]}, {k: units[k] / n[k] for k in n if n[k]})

    def latency(self, lo, hi, n=256):
        """single-batch latency: submit -> complete, one batch in flight (BASELINE metric: p99 batch latency)"""
        torch = self.torch
        eng, stream = self.engines[0], self.sstreams[0]
        idx = [s for s in range(lo, hi) if self.seq[s][0] == 0] or [lo]
        lat = []
        for k in range(n):
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            eng.eval_dev(self.batches[idx[k % len(idx)]], self.scratch[0].c)
            b_.record(stream)
            b_.synchronize()
            lat.append(a.elapsed_time(b_) * 1e3)
        lat.sort()
        return {"unit": "us", "p50": round(percentile(lat, 0.5), 2), "p99": round(percentile(lat, 0.99), 2), "min": round(lat[0], 2),
                "n": len(lat), "what": f"one {self.ctx.B}-request batch, HIP events around guber_eval_batch_dev, nothing else in flight"}

    def host_batches(self, lo, hi):
        return [self.streams.bench_batch(self.ctx.table, self.seq[s][1], self.seq[s][2], algorithm=self.algo_id) for s in range(lo, hi)]

    def close(self):
        if self.workers is not None:
            self.workers.close()
        for e_ in self.engines:
            e_.close()
        self.keep.clear()
        self.batches = []


def parity_gate(rig, orc, threads, now0, label):
    """the GPU's answers for the first kept batches of the rig's sequence against the oracle evaluating the same
    batches in the same order; returns (ok, oracle outputs consumed)"""
    import support
    ctx = rig.ctx
    for lo in range(0, len(ctx.my_ids), 1 << 18):
        orc.eval(rig.streams.bench_batch(ctx.table, ctx.my_ids[lo:lo + (1 << 18)], now0, hits=0, algorithm=rig.algo_id), threads=threads)
    ok = True
    hb = rig.host_batches(0, len(rig.kept))
    for s, b in enumerate(hb):
        want = orc.eval(b, threads=threads)
        try:
            support.assert_results_equal(rig.kept[s].host(), want, f"{label} batch {s}")
        except AssertionError as ex:
            ok = False
            print("PARITY FAILURE:", ex, file=sys.stderr)
    return ok


def cpu_baseline(rig, orc_by_threads, lo, hi, seconds):
    """the oracle in the reference's worker-sharded design (oracle_eval_batch_mt: XXH64-range sharding to W worker caches,
    one thread per worker) on this host's cores, same resident keys, same stream, a time-bounded sample per W"""
    hb = rig.host_batches(lo, hi)
    res = {}
    for w, orc in orc_by_threads.items():
        done, t0 = 0, time.perf_counter()
        while True:
            orc.eval(hb[done % len(hb)], threads=(w if w > 1 else 0))
            done += 1
            el = time.perf_counter() - t0
            if el >= seconds:                      # the sample is bounded by time: the stream's batches are replayed until it is up
                break
        res[w] = (done * rig.ctx.B / el, done, el)
    return res


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import gubernator_amd as ga
    import streams
    from gubernator_amd import shard

    ctx = Ctx()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    red_dev = dev if args.backend == "nccl" else None
    ctx.world, ctx.rank, ctx.local_rank, ctx.dev = world, rank, local_rank, dev
    ctx.K, ctx.B = args.keys, args.batch
    ctx.dispatch = args.dispatch
    ctx.streams = args.streams
    ctx.router = args.router or ("slots" if args.dispatch == "one" else "ring")
    ctx.barrier = (lambda: dist.barrier()) if world > 1 else (lambda: None)
    ctx.max_over_ranks = lambda v: shard.max_over_ranks(v, device=red_dev)
    K, B = args.keys, args.batch
    GSYNC = args.global_sync
    NOW0 = streams.NOW0

    def route_on_device(eng, ring, kb, ko):       # ReplicatedConsistentHash.Get for a chunk of keys (k_route)
        d_kb, d_ko = torch.from_numpy(kb).to(dev), torch.from_numpy(ko.view(np.int32)).to(dev)
        d_owner = torch.empty(len(ko) - 1, dtype=torch.int32, device=dev)
        eng.route_dev(ring, d_kb.data_ptr(), d_ko.data_ptr(), len(ko) - 1, d_owner.data_ptr())
        return d_owner.cpu().numpy()
    ctx.route_on_device = route_on_device

    # ---- key ownership: ids of the global key space (world x K) this rank owns on the ring ------
    total_keys = K if GSYNC else K * world      # GLOBAL: one key space, replicated on every GPU
    ctx.table = streams.key_table(total_keys)
    if GSYNC or world == 1:
        ctx.my_ids = np.arange(total_keys, dtype=np.int64)
    else:
        tmp = ga.Engine(cache_size=1024, device=local_rank, max_batch=1024)
        ctx.my_ids = shard.owned_key_ids(ctx.table, world, rank, route=lambda ring, kb, ko: route_on_device(tmp, ring, kb, ko), chunk=4_000_000)
        tmp.close()

    if GSYNC:
        return run_global(args, ctx, dist)

    S = max(1, args.shards)
    seed = 1234 + rank * 64
    rig = Rig(ctx, args.algo, args.dist, S)
    resident = rig.populate(NOW0)
    m = rig.measure(args.steps, args.warmup, args.min_ms, NOW0, seed)
    lo, hi = args.warmup, args.warmup + args.steps

    roofline = latency = cpu = parity = None
    extras = {}
    if rank == 0:
        # ---- per-kernel durations (HIP events on the engine stream around every launch), one batch in flight ----
        fused = args.dispatch == "one" and S > 1
        per_launch = {}
        if args.profile_steps <= 0:
            kernel_ms = {}
        elif fused:
            kernel_ms, per_launch = rig.kernel_profile_routed(lo, hi)
        else:
            kernel_ms = rig.kernel_profile(args.profile_steps, lo, hi)
        latency = rig.latency(lo, hi)
        cand = {k: v for k, v in kernel_ms.items() if k in KERNEL_BYTES[args.algo] and v > 0}
        if cand:
            # the dominant kernel = the one the GPU spends most time in; its bytes per launch = bytes per request x the
            # requests one launch carries (a fused launch carries the batches of up to four shards)
            dom = max(cand, key=cand.get)
            dom_bytes = int(KERNEL_BYTES[args.algo][dom] * per_launch.get(dom, B))
            achieved = dom_bytes / (cand[dom] * 1e-3) / 1e9
            traffic = measured = None
            tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(args.algo, {}).get(dom.replace("_multi", ""))        # PMC bytes per 65536-request launch (profiles/)
                if traffic and dom in per_launch:
                    traffic = int(traffic * per_launch[dom] / B)
                hp = json.load(open(os.path.join(ROOT, "profiles", "hbm_peak.json")))
                cor = sum(tj.get(args.algo, {}).get(k, 0) for k in ("k_front", "k_eval2"))
                raw = sum(tj.get(args.algo + "_raw", {}).get(k, 0) for k in ("k_front", "k_eval2"))
                if cor and raw:
                    g_raw, g_cor = (x / (m["ms_per_step"] * 1e-3) / 1e9 for x in (raw, cor))
                    measured = {"stream_read_GBps": hp["stream_read_GBps"], "random_gather_GBps": hp["random_gather_128B_GBps"],
                                "hbm_traffic_bytes_per_batch": {"raw": raw, "corrected": cor, "source": tj.get("source", "profiles/roofline_traffic.json")},
                                "hbm_traffic_GBps": {"raw": round(g_raw, 1), "corrected": round(g_cor, 1)},
                                "frac_of_random_gather": {"raw": round(g_raw / hp["random_gather_128B_GBps"], 4),
                                                          "corrected": round(g_cor / hp["random_gather_128B_GBps"], 4)}}
            except Exception:   # noqa: BLE001
                pass
            pipe = BYTES_PER_DECISION[args.algo] * B / (m["ms_per_step"] * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic,
                        "algorithmic_bytes_per_launch": dom_bytes,
                        "kernel_avg_us": {k: round(v * 1e3, 2) for k, v in kernel_ms.items() if v > 0},
                        "requests_per_launch": round(per_launch.get(dom, B), 1),
                        "kernel_timing": ("HIP events around every launch on the shards' stream while the timed steps run once more, all shards "
                                          "overlapping: a launch carries the next batch of up to four shards (k_front_multi / k_eval2_multi)") if fused
                        else "HIP events around every launch on the engine stream, one batch in flight",
                        "measured_ceilings": measured,
                        "pipeline": {"bytes_per_decision": BYTES_PER_DECISION[args.algo],
                                     "ms_per_batch": round(m["ms_per_step"], 5),
                                     "achieved": round(pipe, 2), "frac": round(pipe / HBM_PEAK_GBPS, 6),
                                     "what": "algorithmic bytes of the whole pipeline / timed ms per step (all shards overlapping)"}}

    # ---- CPU baseline + parity gate (rank 0, N = 1 only) -------------------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import support
        ncpu = os.cpu_count() or 1
        ws = []
        for tok in args.cpu_threads.split(","):
            tok = tok.strip()
            if tok:
                w = ncpu if tok == "all" else max(1, min(ncpu, int(tok)))
                if w not in ws:
                    ws.append(w)
        orcs = {w: support.Oracle(cache_size=4 * K, workers=w) for w in ws}
        gate_w = max(ws)
        ok = parity_gate(rig, orcs[gate_w], gate_w, NOW0, "headline")
        for w in ws:
            if w != gate_w:
                for s in range(0, len(ctx.my_ids), 1 << 18):
                    orcs[w].eval(streams.bench_batch(ctx.table, ctx.my_ids[s:s + (1 << 18)], NOW0, hits=0, algorithm=rig.algo_id), threads=(w if w > 1 else 0))
        parity = f"bit-exact vs oracle on the first {len(rig.kept)} batches of the timed stream" if ok else "FAILED"
        if not ok:
            raise SystemExit("parity gate failed: refusing to report a number")
        res = cpu_baseline(rig, orcs, len(rig.kept), min(len(rig.seq), len(rig.kept) + 64), args.cpu_seconds)
        best = max(res, key=lambda w: res[w][0])
        cpu = {"value": round(res[best][0], 1), "unit": "decisions/s", "cores": best, "kind": "port",
               "sample": f"{res[best][1]} batches of {B} from the same routed stream ({res[best][2]:.1f} s), {K} resident keys, the oracle in the "
                         f"reference's worker-sharded design (W caches / threads, XXH64-range sharding, workers.go:180-184); host has {ncpu} cores",
               "by_threads": {str(w): {"value": round(v[0], 1), "batches": v[1], "seconds": round(v[2], 2)} for w, v in res.items()}}
        for o in orcs.values():
            o.close()

    headline_cfg = {"workload": f"{K} resident keys per GPU, one {args.dist} request stream" + (" s=1.1" if args.dist == "zipf" else "") +
                                f" over them, batch={B}, {args.algo.upper()}_BUCKET, hits=1 limit=100 duration=60000ms, {world}xMI355X"
                                + (", keys sharded by replicated consistent hash (512 vnodes, fnv1)" if world > 1 else "")
                                + ((f", {S} logical shards per GPU (own table each; the stream is routed key by key, a shard flushes a batch "
                                    f"when {B} requests are waiting), " +
                                    ("own stream + batcher thread each" if args.dispatch == "threads" else
                                     f"one dispatcher, shards spread over {args.streams} stream(s): the next batch of up to four shards of a stream per pair of launches")) if S > 1 else ", one table"),
                    "keys_per_gpu": K, "batch": B, "algorithm": args.algo, "resident_items_rank0": int(resident),
                    "logical_shards_per_gpu": S, "dispatch": args.dispatch if S > 1 else "caller thread", "placement": rig.placement, "host_cores": os.cpu_count()}
    rig.close()
    del rig
    torch.cuda.empty_cache()

    # ---- extras (rank 0 of a 1-GPU run): other configurations, same machinery, shorter timed region ----
    if rank == 0 and world == 1 and args.extras:
        want = [x.strip() for x in args.extras.split(",") if x.strip()]
        for name in want:
            try:
                extras[name] = run_extra(name, args, ctx, NOW0, seed)
            except Exception as ex:   # noqa: BLE001
                extras[name] = {"error": repr(ex)}
            torch.cuda.empty_cache()

    if rank == 0:
        out = {
            "metric": "rate-limit decisions/sec (kernel path, inputs resident in HBM)",
            "value": round(m["value"], 1), "unit": "decisions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(m["ms_per_step"], 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "int64" if args.algo == "token" else "f64", "data": "synthetic",
            "config": headline_cfg,
            "timed_region": {"repeats_of_the_step_list": m["repeats"], "ms": round(m["timed_ms"], 2), "min_ms": args.min_ms,
                             "ms_per_step_hip_events": round(m["ms_per_step_events"], 5),
                             "enqueue": ("caller thread" if S == 1 else f"{S} pre-started batcher threads behind a barrier" if args.dispatch == "threads"
                                         else f"one dispatcher for {S} shards over {args.streams} stream(s) (guber_eval_batches_routed_dev: batches of shards that share a stream share launches)"),
                             "shard_streams": m["shard_streams"]},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "batch_latency": latency,
        }
        out.update(extras)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def run_extra(name, args, ctx, NOW0, seed):
    """one extra configuration -> sub-object of the JSON line"""
    import support
    K, B = ctx.K, ctx.B
    steps = max(8, min(args.steps, 64))
    warmup = max(4, min(args.warmup, 16))
    min_ms = min(args.min_ms, 150.0)
    if name == "end_to_end":
        return run_end_to_end(args, ctx, NOW0, seed)
    if name == "pool":
        return run_pool(args)
    algo, dist_kind, S = {"leaky": ("leaky", "zipf", max(1, args.shards)), "shards_1": (args.algo, args.dist, 1),
                          "uniform": (args.algo, "uniform", max(1, args.shards))}[name]
    rig = Rig(ctx, algo, dist_kind, S)
    rig.populate(NOW0)
    m = rig.measure(steps, warmup, min_ms, NOW0, seed)
    out = {"value": round(m["value"], 1), "unit": "decisions/s", "ms_per_step": round(m["ms_per_step"], 5), "steps": steps,
           "repeats": m["repeats"], "dtype": "int64" if algo == "token" else "f64",
           "workload": f"{K} keys, {dist_kind}, {algo.upper()}_BUCKET, batch {B}, {S} logical shard(s)"}
    if name in ("leaky", "shards_1"):
        out["batch_latency"] = rig.latency(warmup, warmup + steps, n=128)
        km = rig.kernel_profile(16, warmup, warmup + steps)
        out["kernel_avg_us"] = {k: round(v * 1e3, 2) for k, v in km.items() if v > 0}
    if name == "leaky" and not args.no_cpu_baseline:
        w = min(os.cpu_count() or 1, 32)
        orc = support.Oracle(cache_size=4 * K, workers=w)
        ok = parity_gate(rig, orc, w, NOW0, "leaky")
        orc.close()
        out["parity"] = f"bit-exact vs oracle on the first {len(rig.kept)} batches (tolerance 0)" if ok else "FAILED"
        if not ok:
            out.pop("value")
    rig.close()
    return out


def run_pool(args):
    """the drop-in surface: caller threads -> V1Instance::GetRateLimits -> GPUWorkerPool (C++), tools/bench_pool.cpp"""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "bench_pool_c")
    if not os.path.exists(exe):
        return {"error": "tools/bench_pool_c is not built (make -C gubernator_amd/csrc bench_pool)"}
    T, S, items, keys = 64, 4, 1000, 1_000_000
    p = subprocess.run([exe, str(T), str(S), str(items), str(keys), "1.5"], capture_output=True, text=True, timeout=120)
    m = re.search(r"([0-9.]+) M decisions/s,\s+([0-9.]+) batches/s, avg batch\s+([0-9.]+) requests, errors (\d+)", p.stdout)
    if not m:
        return {"error": (p.stdout + p.stderr)[-400:]}
    return {"value": float(m.group(1)) * 1e6, "unit": "decisions/s", "batches_per_s": float(m.group(2)), "avg_batch": float(m.group(3)),
            "errors": int(m.group(4)),
            "workload": f"{T} caller threads x RPCs of {items} requests through V1Instance::GetRateLimits -> GPUWorkerPool ({S} shards, batch limit 65536, "
                        f"batch wait 200 us), {keys} keys Zipf-1.1, closed loop: validation, HashKey, shard routing, slot reservation, in-place stage "
                        "filling, submit / wait and response fan-out included (what the Go shim does per request)"}


def run_end_to_end(args, ctx, NOW0, seed):
    """host memory in, host memory out: guber_stage_* — requests written into device-visible host arrays, the kernels read them
    and write the responses in place over PCIe, two batches in flight (one being filled / drained by the host while the other
    is evaluated).  Reported with and without the host's fill copy."""
    import gubernator_amd as ga
    import streams
    K, B = ctx.K, ctx.B
    rig = Rig(ctx, args.algo, args.dist, 1)
    rig.populate(NOW0)
    NB = 288
    seq = rig.build_sequence(NB, NOW0, seed)
    eng = rig.engines[0]
    hb = [streams.bench_batch(ctx.table, ids, now, algorithm=rig.algo_id) for (_, ids, now) in seq]
    depth = max(1, int(os.environ.get("GUBER_BENCH_E2E_DEPTH", "3")))      # batches in flight
    NS = depth + 1                                                          # one more stage is being filled / drained by the host
    stages = [ga.Stage(eng, B, key_bytes_cap=B * 16) for _ in range(NS)]
    for st in stages:
        st.disable("burst", "created_at", "is_owner")
    out = {}
    for label, fill in (("with_host_fill", True), ("prefilled", False)):
        if not fill:
            for k, st in enumerate(stages):
                st.fill(hb[k])
        t_sub, lat = {}, []
        h_submit = h_wait = 0.0
        t0 = time.perf_counter()
        for i in range(NB):
            st = stages[i % NS]
            if fill:
                st.fill(hb[i])
            t_sub[i] = time.perf_counter()
            st.submit()
            h_submit += time.perf_counter() - t_sub[i]
            if i >= depth - 1:
                j = i - (depth - 1)
                tw = time.perf_counter()
                stages[j % NS].wait()
                h_wait += time.perf_counter() - tw
                lat.append((time.perf_counter() - t_sub[j]) * 1e6)
        for j in range(NB - depth + 1, NB):
            stages[j % NS].wait()
            lat.append((time.perf_counter() - t_sub[j]) * 1e6)
        el = time.perf_counter() - t0
        lat = sorted(lat[32:])
        bytes_per_req = (15 + 4 + 3 * 8 + 1 + 4) + 26              # every request column present crosses PCIe once, responses once
        out[label] = {"value": round(NB * B / el, 1), "ms_per_step": round(el / NB * 1e3, 4),
                      "latency_us": {"p50": round(percentile(lat, 0.5), 1), "p99": round(percentile(lat, 0.99), 1), "n": len(lat)},
                      "pcie_GBps": round(bytes_per_req * B * NB / el / 1e9, 2),
                      "host_us_per_batch": {"submit": round(h_submit / NB * 1e6, 1), "wait": round(h_wait / NB * 1e6, 1)}}
    # parity of the path: the last prefilled pass must equal what the oracle-checked device path gives — checked in tests/ (test_stage_*)
    for st in stages:
        st.close()
    rig.close()
    best = out["prefilled"]
    return {"value": best["value"], "unit": "decisions/s", "ms_per_step": best["ms_per_step"], "steps": NB,
            "latency_us": best["latency_us"], "pcie_GBps": best["pcie_GBps"], "host_us_per_batch": best["host_us_per_batch"],
            "with_host_fill": out["with_host_fill"],
            "workload": f"{K} keys, {args.dist}, {args.algo.upper()}_BUCKET, batch {B}, one table, guber_stage_submit / guber_stage_wait from one host thread: "
                        "request columns -> the stage's HBM mirror by DMA on a copy stream beside the previous batches' kernels, responses written "
                        f"straight into the host arrays by k_eval2 (GUBER_NO_STAGE_DMA=1: everything read / written in place), {depth} batches in flight; "
                        "`with_host_fill` adds the copy of every batch into the stage (numpy, one thread)"}


def run_global(args, ctx, dist):
    """BASELINE config 5: every request carries GLOBAL, every rank serves all keys from its replica, every K steps the
    ranks exchange pending hits / owner state."""
    import torch
    import gubernator_amd as ga
    import streams
    from gubernator_amd import shard
    K, B, world, rank, dev = ctx.K, ctx.B, ctx.world, ctx.rank, ctx.dev
    GSYNC, NOW0 = args.global_sync, streams.NOW0
    rig = Rig(ctx, args.algo, args.dist, 1, flags=ga.FLAG_GLOBAL, max_key_bytes=64)
    eng, stream = rig.engines[0], rig.sstreams[0]
    ring = ga.Ring(shard.peer_names(world), 512, "fnv1")
    resident = rig.populate(NOW0)
    total_steps = args.warmup + args.steps
    seq = rig.build_sequence(total_steps, NOW0, 1234 + rank * 64)
    rig.seq = seq
    batches, owners = [], []
    for (_, ids, now) in seq:
        owner_ptr = None
        b = rig.dev_batch(ids, now)
        if world > 1:                 # is_owner[i] = (ring owner of key i == this rank), on device
            d_owner = torch.empty(len(ids), dtype=torch.int32, device=dev)
            eng.route_dev(ring, b.key_bytes, b.key_off, len(ids), d_owner.data_ptr())
            t = (d_owner == rank).to(torch.uint8)
            owners.append(t)
            b.is_owner = t.data_ptr()
        batches.append(b)
    eng.max_batch = B
    if args.global_host:      # host-staged exchange (numpy rows, pickled all_gather): kept for comparison
        from gubernator_amd import global_sync
        transport = global_sync.TorchTransport() if world > 1 else type("T", (), {"all_gather": staticmethod(lambda o: [o])})()
        gsync = global_sync.GlobalSync(eng, rank, world, ring, transport)
    else:                     # rows stay in HBM: take_dev -> route -> RCCL all_to_all / all_gather -> eval_dev / add_items_dev
        from gubernator_amd import global_sync_dev
        if world > 1:
            transport = global_sync_dev.TorchTransportDev(dev)
        else:
            transport = type("T", (), {"exchange_rows": staticmethod(lambda send, counts: send),
                                       "gather_rows": staticmethod(lambda rows: [rows])})()
        gsync = global_sync_dev.GlobalSyncDev(eng, rank, world, ring, transport, dev, key_stride=64)
    sync_stats = []

    def run(s):
        eng.eval_dev(batches[s], rig.scratch[0].c)
        if (s + 1) % GSYNC == 0:
            t_s = time.perf_counter()
            with torch.cuda.stream(stream):          # the engine's stream: torch ops of the exchange and engine kernels stay ordered
                st = gsync.sync(NOW0 + 1 + s)
                stream.synchronize()
            st["ms"] = (time.perf_counter() - t_s) * 1e3
            sync_stats.append(st)
    for s in range(args.warmup):
        run(s)
    torch.cuda.synchronize(dev)
    ctx.barrier()
    t0 = time.perf_counter()
    for s in range(args.warmup, total_steps):
        run(s)
    torch.cuda.synchronize(dev)
    wall = ctx.max_over_ranks(time.perf_counter() - t0)
    ctx.barrier()
    if rank == 0:
        timed = sync_stats[args.warmup // GSYNC:] or sync_stats
        out = {"metric": "rate-limit decisions/sec (kernel path, inputs resident in HBM)", "value": round(args.steps * B * world / wall, 1),
               "unit": "decisions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(wall / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "int64" if args.algo == "token" else "f64", "data": "synthetic",
               "config": {"workload": f"{K} keys replicated on every GPU, {args.dist} stream, batch={B}, {args.algo.upper()}_BUCKET, GLOBAL behaviour, "
                                      f"sync every {GSYNC} batches, {world}xMI355X", "keys_per_gpu": K, "batch": B, "resident_items_rank0": int(resident)},
               "global_sync": {"every_batches": GSYNC, "syncs": len(sync_stats),
                               "avg_ms": round(sum(x["ms"] for x in timed) / max(len(timed), 1), 3),
                               "avg_rows_broadcast": int(sum(x["broadcast"] for x in timed) / max(len(timed), 1)),
                               "avg_hits_rows_sent": int(sum(x["hits_sent"] for x in timed) / max(len(timed), 1)),
                               "bytes_moved_rank0": gsync.bytes_moved,
                               "exchange": "host-staged" if args.global_host else "device-resident (RCCL on HBM rows)"}}
        print(json.dumps(out))
    rig.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
