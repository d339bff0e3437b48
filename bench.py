#!/usr/bin/env python3
"""bench.py — rate-limit decisions/sec of the MI355X engine on BASELINE.json's workload.

One "step" = one GetRateLimits batch of 65536 checks evaluated by the HIP path (k_front -> k_eval2, or their fused
multi-table forms) with every input array already resident in HBM.  Workload (BASELINE.json configs[1], SURVEY.md section
8d): 10M resident keys per GPU, ONE Zipf(1.1) request stream over those keys (numpy PCG64 stream seed 1234, permutation
seed 99), TOKEN_BUCKET, hits 1, limit 100, duration 60 s, now_ms advancing 1 ms per batch.

The stream is NEVER replayed: the timed region is max(--steps, 2048) DISTINCT batches drawn once (134 M requests touching
~5.6 M distinct keys = ~0.8 GB of table lines at 144 B per key, 1.1 GB in 64-byte sectors — past the 256 MiB Infinity
Cache), each with its own now_ms.  Warm-up batches, the batches used for per-kernel HIP-event timing and for the
single-batch latency are further distinct parts of the same stream.  Key ids are drawn on the host (the canonical numpy
generator), ranked on the device (torch.searchsorted over the exact CDF), routed to the shards and gathered into key bytes by
device-side index ops — all outside the clock.

Inside a GPU the resident keys are split into S logical shards (default 12; the reference shards its key space the same way
over Config.Workers goroutines, workers.go:19-25,125-151): S engines with their own HBM tables.  Placement of keys on shards
is the product's (guber_placement_*, gubernator_amd/csrc/placement.cpp: hash slots + individually placed hot key HASHES,
fitted to a 2 M-request sample of earlier traffic; --router plain = untouched XXH64 ranges = the reference's getWorker).  The
stream is split request by request by that placement, and a shard flushes a batch whenever 65536 of its requests are waiting
— the policy of the reference's batcher (peer_client.go:284-337) — so batches are exactly 65536 requests, hot shards flush
more often, and per-key request order is the stream's order.  The per-request routing itself is the front end's work (the
callers of GPUWorkerPool do it while they write their requests into the shard's stage) and is NOT in the kernel-path number:
`shards_1` is the literal single-table configuration and `pool` the full product surface, routing included.

Timing: all timed batches are enqueued by ONE dispatcher call (guber_eval_batches_routed_dev: round by round the next batch
of every shard; shards that share a stream — 12 shards over 3 streams by default — share their two launches), bracketed by
barrier + device synchronize; nothing is created or allocated inside.  Parity is checked over the TIMED work: the oracle
is fed the whole stream (populate, warm-up, every timed batch, in order) and every 64th timed batch's answers (plus the
first 8 and the last) must equal the oracle's element-wise; the engine's internal-retry counter must not have moved.

Extras in the same JSON line: `leaky` (configs[2], parity-gated), `expiring` (duration 500 ms: every bucket expires and is
renewed several times under the clock), `shards_1`, `uniform` (no skew), `end_to_end` (host memory in / out, PCIe
included), `pool` (caller threads -> V1Instance::GetRateLimits -> the C++ GPUWorkerPool, 10 M keys).

N > 1 (launched by torch.distributed.run, one rank per GPU): the key space is N x 10M keys sharded by the reference's
replicated consistent hash (replicated_hash.go; 512 vnodes, fnv1, peers gpu0..gpuN-1), every rank evaluates the requests
for the keys it owns — no data-path collective (weak scaling).  --global-sync K = BASELINE config 5 on the native exchange
(guber_comm_* + guber_global_sync: RCCL between ranks, device copies between logical ranks of one GPU).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0           # MI355X spec (MI355X_MICROARCH.md)
MIN_TIMED_BATCHES = 2048         # the timed region is never shorter than this many distinct batches
BYTES_PER_DECISION = {"token": 149, "leaky": 173}   # SURVEY.md section 8d, 16-byte keys
# split of the algorithmic bytes over the kernels that touch request / table / response data (DESIGN.md
# "Algorithmic bytes"): k_front reads key_off 4 + key 16 + table 56 (token) / 64 (leaky); k_eval2 reads
# the request fields 32 / 40, writes table 16 / 24 and the response 25.
KERNEL_BYTES = {"token": {"k_front": 76, "k_eval2": 73, "k_front_multi": 76, "k_eval2_multi": 73,
                          # owner-partitioned pipeline: k_part reads key_off 4 + key 16, k_own the table 56, k_eval3 the request fields 32,
                          # writes table 16 + response 25
                          "k_part": 20, "k_own": 56, "k_eval3": 73, "k_part_multi": 20, "k_own_multi": 56, "k_eval3_multi": 73,
                          "k_evalpart_multi": 93},     # (GUBER_FUSE_EP: one batch's k_eval3 + the next one's k_part in one launch)
                "leaky": {"k_front": 84, "k_eval2": 89, "k_front_multi": 84, "k_eval2_multi": 89,
                          "k_part": 20, "k_own": 64, "k_eval3": 89, "k_part_multi": 20, "k_own_multi": 64, "k_eval3_multi": 89,
                          "k_evalpart_multi": 109}}


DIGEST_C = (-7046029254386353131, -4417276706812531889, 1609587929392839161, -8796714831421723037)   # odd 64-bit multipliers (int64 view)


def host_digest(res):
    """Rig.result_digests() for one HostResult, in numpy (uint64 wrap-around)"""
    u = lambda a: np.asarray(a).astype(np.int64).view(np.uint64)
    cc = [np.uint64(x & 0xffffffffffffffff) for x in DIGEST_C]
    n = len(res.status)
    with np.errstate(over="ignore"):
        v = (np.asarray(res.status).astype(np.uint64) | (np.asarray(res.err).astype(np.uint64) << np.uint64(8))) * cc[0]
        v = v ^ (u(res.limit) * cc[1]) ^ (u(res.remaining) * cc[2]) ^ (u(res.reset_time) * cc[3])
        idx = np.arange(n, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
        return int((v * idx).sum(dtype=np.uint64))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048, help=f"timed batches; at least {MIN_TIMED_BATCHES} distinct batches are always timed")
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--keys", type=int, default=10_000_000, help="resident keys per GPU")
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--algo", choices=["token", "leaky"], default="token")
    ap.add_argument("--dist", choices=["zipf", "uniform"], default="zipf")
    ap.add_argument("--duration-ms", type=int, default=60_000, help="RateLimitReq.duration of the stream")
    ap.add_argument("--min-batches", type=int, default=MIN_TIMED_BATCHES, help="lower bound of the timed region in distinct batches (measurement scripts may lower it)")
    ap.add_argument("--min-ms", type=float, default=0.0, help="accepted and ignored (the timed region is a fixed number of distinct batches, never a replay)")
    ap.add_argument("--gen-batches", type=int, default=16, help="routed: batches of --batch requests per generation handed to the device-side front (guber_front_*)")
    ap.add_argument("--headline", choices=["routed", "presplit"], default="routed",
                    help="routed (the all-inclusive arrangement): ONE raw request stream in arrival order -> routed to the shards on the device -> answers in request "
                         "order, inside the clock; presplit: rounds 2-5's headline (per-shard batches split outside the clock, answers left in shard order)")
    ap.add_argument("--extras", default="presplit,routed_leaky,leaky,expiring,shards_1,uniform,end_to_end,pool,global_sync,two_ranks",
                    help="comma list of extra configurations measured after the headline one (N = 1 only); '' = none")
    ap.add_argument("--extra-batches", type=int, default=1024, help="timed distinct batches of the extra configurations")
    ap.add_argument("--dispatch", choices=["threads", "one"], default="one",
                    help="who enqueues the shards' batches: one pre-started thread per shard, or ONE dispatcher for all shards in flush order "
                         "(guber_eval_batches_routed_dev)")
    ap.add_argument("--router", choices=["placed", "plain"], default="placed",
                    help="placement of a GPU's keys on its logical shards: guber_placement fitted to a sample of earlier traffic, or its "
                         "untouched initial table (XXH64 ranges = the reference's getWorker)")
    ap.add_argument("--streams", type=int, default=3, help="with --dispatch one: streams the shards are spread over (shards of one stream share launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline AND the oracle parity pass")
    ap.add_argument("--cpu-seconds", type=float, default=3.0, help="CPU time budget per thread count of the baseline")
    ap.add_argument("--cpu-threads", default="1,8,16,64,all", help="worker shards / threads of the CPU baseline (comma list, 'all' = every host core)")
    ap.add_argument("--profile-steps", type=int, default=4096,
                    help="further distinct batches run once more, dispatched like the timed region, with HIP events around every launch: the per-kernel "
                         "durations and the batch latency under load (4096 batches = >= 1000 pipeline passes of up to four tables)")
    ap.add_argument("--latency-steps", type=int, default=1024, help="distinct batches run one at a time for the single-batch latency")
    ap.add_argument("--shards", type=int, default=12, metavar="S",
                    help="logical key-space shards per GPU (the reference's Config.Workers sharding, workers.go:19-25): S "
                         "engines with their own tables, the stream routed to them key by key")
    ap.add_argument("--global-sync", type=int, default=0, metavar="K",
                    help="BASELINE config 5: every request carries GLOBAL, every rank serves ALL keys from its replica, "
                         "and every K steps the ranks run guber_global_sync (0 = off)")
    ap.add_argument("--logical-ranks", type=int, default=2, help="with --global-sync on ONE process: logical ranks sharing the GPU (device-copy transport)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL)")
    ap.add_argument("--ring-peers", type=int, default=0,
                    help="N > 1: peers on the consistent-hash ring (default = --gpus).  More peers than ranks = BASELINE config 4's shape rehearsed with fewer "
                         "GPUs: the key space is ring-peers x --keys, rank r owns what the ring gives gpu<r>")
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


class ZipfRanker:
    """The stream of tests/streams.py ZipfSampler (same generator, same seeds, same values), drawn in bulk: the uniforms
    come from the canonical numpy generator, the inverse-CDF search and the rank -> key permutation run on the device."""

    def __init__(self, torch, dev, n_keys, s=1.1, seed=1234, perm_seed=99):
        w = 1.0 / np.power(np.arange(1, n_keys + 1, dtype=np.float64), s)
        cdf = np.cumsum(w)
        cdf /= cdf[-1]
        self.torch, self.dev, self.n_keys = torch, dev, n_keys
        self.d_cdf = torch.from_numpy(cdf).to(dev)
        self.d_perm = torch.from_numpy(np.random.default_rng(perm_seed).permutation(n_keys).astype(np.int32)).to(dev)
        self.rng = np.random.default_rng(seed)

    def draw_dev(self, n, chunk=1 << 25):
        """-> int32 device tensor of n local key ids"""
        torch = self.torch
        out = torch.empty(n, dtype=torch.int32, device=self.dev)
        for lo in range(0, n, chunk):
            m = min(chunk, n - lo)
            u = torch.from_numpy(self.rng.random(m)).to(self.dev)
            r = torch.searchsorted(self.d_cdf, u, right=False).clamp_(max=self.n_keys - 1)
            out[lo:lo + m] = self.d_perm[r]
        return out


def split_stream(torch, d_ids, d_sown, S, B, total):
    """One request stream (d_ids: key ids in arrival order) -> the first `total` batches in flush order when every shard
    flushes a batch as soon as B of its requests are waiting (peer_client.go:284-337): (ids[total, B], shard of every batch).
    Inside a batch and between the batches of one shard the stream's order is kept."""
    if S <= 1:
        if d_ids.numel() < total * B:
            raise SystemExit("stream too short for the requested number of batches")
        return d_ids[:total * B].view(total, B), [0] * total
    d_sh = d_sown.index_select(0, d_ids)
    order = torch.sort(d_sh, stable=True).indices                    # stream positions grouped by shard, ascending inside
    counts = torch.bincount(d_sh.to(torch.int32), minlength=S).cpu().numpy()
    del d_sh
    rows, meta, start = [], [], 0
    for j in range(S):
        nb = int(counts[j]) // B
        if nb:
            pos = order[start:start + nb * B].view(nb, B)
            rows.append(pos)
            last = pos[:, B - 1].cpu().numpy()
            meta += [(int(last[k]), j) for k in range(nb)]           # a batch is flushed when its last request arrives
        start += int(counts[j])
    idx = sorted(range(len(meta)), key=lambda q: meta[q][0])[:total]
    if len(idx) < total:
        raise SystemExit("stream too short for the requested number of batches")
    # a batch flushed at position p is complete only if the stream was drawn at least up to p: true for every batch kept
    sel = torch.cat(rows, 0).index_select(0, torch.tensor(idx, device=d_ids.device))
    return d_ids.index_select(0, sel.reshape(-1)).view(total, B), [meta[q][1] for q in idx]


class Rig:
    """One measured configuration: S engines (tables + streams) over this rank's keys, a routed non-repeating request stream
    resident in HBM, and the machinery to run it."""

    def __init__(self, ctx, algo, dist_kind, S, flags=0, max_key_bytes=0, duration_ms=60_000):
        import torch
        import gubernator_amd as ga
        import streams
        self.ctx, self.algo, self.dist_kind, self.S = ctx, algo, dist_kind, S
        self.duration_ms = int(duration_ms)
        self.dispatch = getattr(ctx, "dispatch", "threads")
        self.algo_id = 0 if algo == "token" else 1
        self.torch, self.ga, self.streams = torch, ga, streams
        dev, B = ctx.dev, ctx.B
        self.sstreams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        if self.dispatch == "one":             # one dispatcher: shards that share a stream share launches
            ns = max(1, min(S, int(getattr(ctx, "streams", 1))))
            self.sstreams = [self.sstreams[j * ns // S] for j in range(S)]
        nk = len(ctx.my_ids)
        # logical shards inside this GPU: the product's placement (guber_placement_*) says which shard holds a key; every shard
        # then gets a table for the keys it really holds (+ 25 %)
        self.placement = None
        self.place = None
        if S > 1:
            self.place = ga.Placement(S)
            info = {"router": "guber_placement: 4096 XXH64-range slots -> shards (initial table = the reference's getWorker, workers.go:180-184)"}
            if getattr(ctx, "router", "placed") == "placed" and dist_kind == "zipf":
                # fitted to what an earlier sample of the traffic carried (2 M requests of the same distribution, another seed)
                obs = streams.ZipfSampler(nk, s=1.1, seed=990_001 + ctx.rank, perm_seed=99).draw(1 << 21)
                self.place.observe_keys(*streams.keys_for_ids(ctx.table, ctx.my_ids[obs]))
                self.place.rebalance(0.125, True)
                info = {"router": "guber_placement (C++, on key hashes): 4096 XXH64-range slots and the keys that alone outweigh 1/8 of a shard's fair "
                                  "share placed longest-processing-time-first on what 2 M earlier requests carried",
                        "keys_placed_individually": self.place.n_hot()}
            sown = np.empty(nk, np.uint8 if S <= 256 else np.uint16)
            for lo in range(0, nk, 2_000_000):
                sh, _ = self.place.route_keys(*streams.keys_for_ids(ctx.table, ctx.my_ids[lo:lo + 2_000_000]))
                sown[lo:lo + len(sh)] = sh
            self.sown = sown
            self.placement = info
        else:
            self.sown = np.zeros(nk, np.uint8)
        self.local_of_shard = [np.nonzero(self.sown == j)[0] for j in range(S)]
        slots_log2 = int(os.environ.get("GUBER_BENCH_TABLE_SLOTS_LOG2", "0"))       # experiments: 0 = the engine's own rule (4 x or 2 x (cache_size + max_batch), rounded up)
        # CacheSize = twice the resident population (the reference's cache must not evict live limits either).  The engine evicts
        # exactly like lrucache.go, which needs to know BEFORE a batch whether it can overflow the cache; the host only knows the
        # exact item count as of the last batch that has reported, plus the requests in flight — so the headroom is also what lets
        # `headroom / batch` batches per shard be in flight without waiting for the GPU (include/guber_gpu.h "bounded cache"): a shard
        # that holds one hot key still sees batches of B requests, hence the 8 batches on top
        self.engines = [ga.Engine(cache_size=2 * len(self.local_of_shard[j]) + 8 * B + 1024, device=ctx.local_rank, max_batch=B,
                                  table_slots=(1 << slots_log2) if slots_log2 else 0,
                                  stream=self.sstreams[j].cuda_stream, max_key_bytes=max_key_bytes, flags=flags) for j in range(S)]
        # arrays every batch of this rig shares (fixed-width keys, constant request fields)
        L = ctx.table.shape[1]
        self.L = L
        self.t_off = torch.from_numpy((np.arange(B + 1, dtype=np.int64) * L).astype(np.int32)).to(dev)
        self.t_hits1 = torch.full((B,), 1, dtype=torch.int64, device=dev)
        self.t_hits0 = torch.zeros((B,), dtype=torch.int64, device=dev)
        self.t_limit = torch.full((B,), 100, dtype=torch.int64, device=dev)
        self.t_dur = torch.full((B,), self.duration_ms, dtype=torch.int64, device=dev)
        self.t_algo = torch.full((B,), self.algo_id, dtype=torch.uint8, device=dev)
        self.t_beh = torch.full((B,), 2 if (flags & ga.FLAG_GLOBAL) else 0, dtype=torch.int32, device=dev)
        self.scratch = [self.DevResult(self, B) for _ in range(S)]
        self.workers = BatcherThreads(S) if (S > 1 and self.dispatch == "threads") else None
        self.keep = []          # tensors referenced by C structs
        self.d_keytab = torch.from_numpy(np.ascontiguousarray(ctx.table[ctx.my_ids])).to(dev)      # (nk, L) key bytes by local id

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

    def batch_struct(self, keys_ptr, n, now_ms, hits=1, owner_ptr=None, row=None):
        """row = index of the batch in the stream: its request columns are its OWN slices of the stream's column tensors (first-touch
        HBM reads, like its keys); None = the rig's shared constant columns (residency pass)"""
        ga = self.ga
        if row is not None:
            c = self.cols
            B = self.ctx.B
            return ga.GuberBatch(n, 0, keys_ptr, c["off"].data_ptr() + row * (B + 1) * 4,
                                 c["hits"].data_ptr() + row * B * 8, c["limit"].data_ptr() + row * B * 8, c["duration"].data_ptr() + row * B * 8,
                                 None, None, c["algorithm"].data_ptr() + row * B, c["behavior"].data_ptr() + row * B * 4, owner_ptr, None, None, int(now_ms))
        return ga.GuberBatch(n, 0, keys_ptr, self.t_off.data_ptr(),
                             (self.t_hits1 if hits else self.t_hits0).data_ptr(), self.t_limit.data_ptr(), self.t_dur.data_ptr(),
                             None, None, self.t_algo.data_ptr(), self.t_beh.data_ptr(), owner_ptr, None, None, int(now_ms))

    def populate(self, now0):
        """residency: every owned key gets a bucket before anything is timed (hits 0 = create, consume nothing)"""
        torch, B, L = self.torch, self.ctx.B, self.L
        pad = torch.zeros(8, dtype=torch.uint8, device=self.ctx.dev)
        for j in range(self.S):
            loc = torch.from_numpy(self.local_of_shard[j]).to(self.ctx.dev)
            for lo in range(0, len(loc), B):
                sel = loc[lo:lo + B]
                kb = torch.cat([self.d_keytab.index_select(0, sel).reshape(-1), pad])
                torch.cuda.current_stream(self.ctx.dev).synchronize()    # kb is produced on torch's stream, consumed on the engine's
                self.engines[j].eval_dev(self.batch_struct(kb.data_ptr(), len(sel), now0, hits=0), self.scratch[j].c)
                self.engines[j].synchronize()
        return sum(e_.size() for e_ in self.engines)

    def build_stream(self, total, now0, seed):
        """Draw ONE request stream over this rank's keys, split it by the placement, flush a shard's batch whenever B of its
        requests are waiting; keep the first `total` batches in flush order.  Everything stays on the device; the local key
        ids of every batch also go to the host for the oracle.  Sets seq = [(shard, now_ms)], d_keys (all batches' key bytes,
        batch s at offset s*B*L), h_ids[(total, B)]."""
        torch, B, S, L, dev = self.torch, self.ctx.B, self.S, self.L, self.ctx.dev
        nk = len(self.ctx.my_ids)
        n = (total + 2 * S + 2) * B
        if self.dist_kind == "zipf":
            d_ids = ZipfRanker(torch, dev, nk, s=1.1, seed=seed, perm_seed=99).draw_dev(n)
        else:
            d_ids = torch.from_numpy(np.random.default_rng(seed).integers(0, nk, n, dtype=np.int32)).to(dev)
        d_sown = torch.from_numpy(self.sown.astype(np.uint8)).to(dev) if S > 1 else None
        d_bids, shard_of = split_stream(torch, d_ids, d_sown, S, B, total)
        del d_ids
        self.seq = [(shard_of[s], now0 + 1 + s) for s in range(total)]
        self.h_ids = d_bids.cpu().numpy()
        d_keys = torch.empty(total * B * L + 8, dtype=torch.uint8, device=dev)
        d_keys[-8:] = 0
        step = 256
        for lo in range(0, total, step):
            hi = min(total, lo + step)
            d_keys[lo * B * L:hi * B * L] = self.d_keytab.index_select(0, d_bids[lo:hi].reshape(-1)).reshape(-1)
        self.d_keys = d_keys
        self.distinct_keys = int(torch.unique(d_bids).numel())
        # table accesses per decision: a batch touches the table once per DISTINCT key (sampled over 16 batches)
        samp = list(range(0, total, max(1, total // 16)))[:16]
        self.access_frac = float(np.mean([int(torch.unique(d_bids[r]).numel()) for r in samp])) / B
        del d_bids
        # every batch reads its OWN request columns (key offsets, hits, limit, duration, behavior, algorithm: 36 B per request with the
        # offset), as it reads its own keys: nothing of a request is served from a cache line an earlier batch left behind
        self.cols = {"off": self.t_off.repeat(total), "hits": torch.full((total * B,), 1, dtype=torch.int64, device=dev),
                     "limit": torch.full((total * B,), 100, dtype=torch.int64, device=dev),
                     "duration": torch.full((total * B,), self.duration_ms, dtype=torch.int64, device=dev),
                     "algorithm": torch.full((total * B,), self.algo_id, dtype=torch.uint8, device=dev),
                     "behavior": self.t_beh.repeat(total)}
        self.batches = [self.batch_struct(d_keys.data_ptr() + s * B * L, B, self.seq[s][1], row=s) for s in range(total)]
        torch.cuda.synchronize(dev)

    def keep_results(self, which):
        """every batch in `which` gets result arrays of its own (slices of five big tensors): its answers stay until they are checked"""
        torch, B, dev = self.torch, self.ctx.B, self.ctx.dev
        which = list(which)
        n = len(which)
        self.res_cols = {"status": torch.empty(n * B, dtype=torch.uint8, device=dev), "err": torch.empty(n * B, dtype=torch.uint8, device=dev),
                         "limit": torch.empty(n * B, dtype=torch.int64, device=dev), "remaining": torch.empty(n * B, dtype=torch.int64, device=dev),
                         "reset_time": torch.empty(n * B, dtype=torch.int64, device=dev)}
        self.kept = {s: self.SliceResult(self, k) for k, s in enumerate(which)}

    class SliceResult:
        def __init__(self, rig, k):
            B, c = rig.ctx.B, rig.res_cols
            self.rig, self.k = rig, k
            self.c = rig.ga.GuberResult(c["status"].data_ptr() + k * B, c["limit"].data_ptr() + k * B * 8, c["remaining"].data_ptr() + k * B * 8,
                                        c["reset_time"].data_ptr() + k * B * 8, c["err"].data_ptr() + k * B, 0, 0, 0, 0, 0)

        def host(self):
            rig, B, k = self.rig, self.rig.ctx.B, self.k
            h = rig.ga.HostResult(B)
            for name in ("status", "limit", "remaining", "reset_time", "err"):
                getattr(h, name)[:] = rig.res_cols[name][k * B:(k + 1) * B].cpu().numpy()
            return h

    def result_digests(self):
        """one 64-bit digest per kept batch, computed on the device over its five result columns (position-dependent, wrap-around
        arithmetic; host_digest() is the same formula in numpy) -> {batch: digest}"""
        torch, B = self.torch, self.ctx.B
        c = self.res_cols
        n = len(self.kept)
        idx = (torch.arange(B, dtype=torch.int64, device=self.ctx.dev) * 2 + 1)
        out = {}
        order = sorted(self.kept, key=lambda s: self.kept[s].k)
        for lo in range(0, n, 64):
            hi = min(n, lo + 64)
            sl = slice(lo * B, hi * B)
            v = (c["status"][sl].to(torch.int64) | (c["err"][sl].to(torch.int64) << 8)) * DIGEST_C[0]
            v = v ^ (c["limit"][sl] * DIGEST_C[1]) ^ (c["remaining"][sl] * DIGEST_C[2]) ^ (c["reset_time"][sl] * DIGEST_C[3])
            v = (v.view(hi - lo, B) * idx).sum(dim=1)
            for k, d in zip(range(lo, hi), v.cpu().numpy().tolist()):
                out[order[k]] = d & 0xffffffffffffffff
        return out

    def _arrays(self, lo, hi):
        """per shard: ctypes arrays (GuberBatch[], GuberResult[], count) of the sequence's batches lo..hi in order"""
        ga = self.ga
        per = []
        for j in range(self.S):
            idx = [s for s in range(lo, hi) if self.seq[s][0] == j]
            ba = (ga.GuberBatch * max(len(idx), 1))(*[self.batches[s] for s in idx])
            ra = (ga.GuberResult * max(len(idx), 1))(*[(self.kept[s].c if s in self.kept else self.scratch[j].c) for s in idx])
            per.append((ba, ra, len(idx)))
        return per

    def _routed(self, lo, hi):
        """the sequence's batches lo..hi in flush order for ONE dispatcher: (which[], GuberBatch[], GuberResult[], count)"""
        ga = self.ga
        idx = list(range(lo, hi))
        wa = (C.c_uint32 * max(len(idx), 1))(*[self.seq[s][0] for s in idx])
        ba = (ga.GuberBatch * max(len(idx), 1))(*[self.batches[s] for s in idx])
        ra = (ga.GuberResult * max(len(idx), 1))(*[(self.kept[s].c if s in self.kept else self.scratch[self.seq[s][0]].c) for s in idx])
        return wa, ba, ra, len(idx)

    def run(self, lo, hi, timed=False):
        """enqueue batches lo..hi of the sequence once.  timed: returns (wall seconds, max per-stream event ms)"""
        torch = self.torch
        one = self.dispatch == "one" and self.S > 1
        if one:
            wa, ba1, ra1, cnt1 = self._routed(lo, hi)
        else:
            per = self._arrays(lo, hi)

            def job(j):
                ba, ra, cnt = per[j]
                eng = self.engines[j]
                return (lambda: eng.eval_many_dev(ba, ra, cnt)) if cnt else None
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
        c0 = time.thread_time()                                       # CPU time of THIS thread (the dispatcher's call runs on it)
        if one:
            self.ga.Engine.eval_routed_dev(self.engines, wa, ba1, ra1, cnt1)
        elif self.workers is not None:
            self.workers.run(jobs)
        else:
            for f in jobs:
                if f:
                    f()
        t_enq = time.perf_counter()
        self.last_enqueue_busy_s = time.thread_time() - c0
        if timed:
            for j in range(self.S):
                ev1[j].record(self.sstreams[j])
        torch.cuda.synchronize(self.ctx.dev)
        t1 = time.perf_counter()
        if timed:
            self.ctx.barrier()
            self.last_stream_ms = [ev0[j].elapsed_time(ev1[j]) for j in range(self.S)]
            self.last_stream_batches = [sum(1 for s in range(lo, hi) if self.seq[s][0] == j) for j in range(self.S)]
            self.last_enqueue_s = t_enq - t0
            return t1 - t0, max(self.last_stream_ms)
        return t1 - t0, None

    def measure(self, steps, warmup, now0, seed, profile_steps=0, latency_steps=0):
        """build the stream, warm up, time `steps` distinct batches once.  -> dict"""
        ctx = self.ctx
        total = warmup + steps + profile_steps + latency_steps
        self.warmup, self.steps, self.profile_steps, self.latency_steps = warmup, steps, profile_steps, latency_steps
        self.build_stream(total, now0, seed)
        lo, hi = warmup, warmup + steps
        self.keep_results(range(lo, hi))                              # every timed batch keeps its answers (checked by digest afterwards)
        retries0 = sum(e.stats()["retries"] for e in self.engines)
        if warmup:
            self.run(0, warmup)
        wall, ev_ms = self.run(lo, hi, timed=True)
        wall = ctx.max_over_ranks(wall)
        self.retries = sum(e.stats()["retries"] for e in self.engines) - retries0
        B = ctx.B
        return {"value": steps * B * ctx.world / wall, "ms_per_step": wall / steps * 1e3, "timed_batches": steps,
                "timed_ms": wall * 1e3, "ms_per_step_events": ev_ms / steps, "enqueue_ms": self.last_enqueue_s * 1e3, "enqueue_busy_ms": self.last_enqueue_busy_s * 1e3,
                "distinct_keys_in_stream": self.distinct_keys,
                "internal_retries": int(self.retries),
                "shard_streams": [{"batches": b, "stream_ms": round(m, 3), "us_per_batch": round(m * 1e3 / max(b, 1), 2)}
                                  for b, m in zip(self.last_stream_batches, self.last_stream_ms)]}

    def kernel_profile(self):
        """the profile segment (distinct batches after the timed ones), dispatched exactly like the timed region, with HIP events
        around every launch.  -> ({kernel: avg ms per launch}, {kernel: avg requests per launch}, {kernel: launches})"""
        lo = self.warmup + self.steps
        hi = lo + self.profile_steps
        if hi <= lo:
            return {}, {}, {}
        for e in self.engines:
            e.profile(True)
            e.profile_read()
        self.run(lo, hi)
        ms, n, units = {}, {}, {}
        self.pass_us = []
        for e in self.engines:
            prof = e.profile_read()
            self.pass_us += e.profile_passes()
            e.profile(False)
            for k, (cnt, tot) in prof.items():
                n[k] = n.get(k, 0) + cnt
                ms[k] = ms.get(k, 0.0) + tot
                units[k] = units.get(k, 0) + e.last_profile_units.get(k, 0)
        return ({k: ms[k] / n[k] for k in n if n[k]}, {k: units[k] / n[k] for k in n if n[k]}, {k: n[k] for k in n if n[k]})

    def latency_under_load(self):
        """what a batch spends on the GPU in the regime the throughput is measured in: the profile segment is dispatched exactly like
        the timed region (every stream busy), and every pipeline pass — the launches of one batch, or of one fused group of up to four
        tables' batches — is bracketed by HIP events on its stream: first kernel's start -> last kernel's end"""
        lat = sorted(getattr(self, "pass_us", []))
        if not lat:
            return None
        return {"unit": "us", "p50": round(percentile(lat, 0.5), 2), "p99": round(percentile(lat, 0.99), 2), "min": round(lat[0], 2), "max": round(lat[-1], 2),
                "n": len(lat), "what": ("first kernel's start -> last kernel's end of every pipeline pass of the profile segment (dispatched like the timed "
                                        "region, all streams busy); a pass carries the next batch of up to four tables, each of which is done when the pass is. "
                                        "Open loop: the whole stream is enqueued at once, so time waiting for a launch slot is the dispatcher's queue, not counted")}

    def latency(self):
        """single-batch latency: submit -> complete, one batch in flight, every batch a fresh one (BASELINE metric: p99 batch latency)"""
        torch = self.torch
        lo = self.warmup + self.steps + self.profile_steps
        hi = lo + self.latency_steps
        lat = []
        for s in range(lo, hi):
            j = self.seq[s][0]
            eng, stream = self.engines[j], self.sstreams[j]
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            eng.eval_dev(self.batches[s], self.scratch[j].c)
            b_.record(stream)
            b_.synchronize()
            lat.append(a.elapsed_time(b_) * 1e3)
        if not lat:
            return None
        lat.sort()
        return {"unit": "us", "p50": round(percentile(lat, 0.5), 2), "p99": round(percentile(lat, 0.99), 2), "min": round(lat[0], 2),
                "n": len(lat), "what": f"one {self.ctx.B}-request batch (each a fresh part of the stream), HIP events around guber_eval_batch_dev, nothing else in flight"}

    def host_batch(self, s):
        ids = self.ctx.my_ids[self.h_ids[s]]
        return self.streams.bench_batch(self.ctx.table, ids, self.seq[s][1], algorithm=self.algo_id, duration=self.duration_ms)

    def close(self):
        if self.workers is not None:
            self.workers.close()
        for e_ in self.engines:
            e_.close()
        if self.place is not None:
            self.place.close()
        self.keep.clear()
        self.batches = []
        self.cols = None
        self.d_keys = None
        self.d_keytab = None
        self.kept = {}


class RoutedRig(Rig):
    """The all-inclusive arrangement (guber_front_*, include/guber_gpu.h): ONE raw request stream in HBM, never split on the host.  A
    generation = `gen_batches` consecutive batches of B requests in ARRIVAL order (what a batcher has collected while the previous
    generation ran: peer_client.go:284-337) -> on the device: XXH64 of every HashKey + the placement's rule -> shard (workers.go:261-289,
    getWorker :180-184), every shard's share contiguous and in arrival order -> the S tables through the fused launches -> the answers
    back in ARRIVAL order (gubernator.proto:51-54) — all inside the clock.  Units: `steps` / `warmup` / seq / kept / host_batch are in
    batches of B (slices of a generation), so the parity and CPU-baseline passes are the headline's, fed in request order."""

    def __init__(self, ctx, algo, dist_kind, S, gen_batches, duration_ms=60_000):
        super().__init__(ctx, algo, dist_kind, S, duration_ms=duration_ms)
        self.GB = int(gen_batches)
        self.G = self.GB * ctx.B
        self.front = self.ga.Front(self.engines, self.place, max_n=self.G, depth=int(os.environ.get("GUBER_BENCH_FRONT_DEPTH", "4")))

    def gen_now(self, g, now0):
        return now0 + 1 + g * self.GB                               # (the clock advances 1 ms per batch of B on average)

    def build_stream(self, total, now0, seed):
        """total = batches of B (a multiple of GB): ONE stream in arrival order, generation g = requests [g*G, (g+1)*G)"""
        torch, B, L, dev, G, GB = self.torch, self.ctx.B, self.L, self.ctx.dev, self.G, self.GB
        assert total % GB == 0
        ngen = total // GB
        nk = len(self.ctx.my_ids)
        n = total * B
        if self.dist_kind == "zipf":
            d_ids = ZipfRanker(torch, dev, nk, s=1.1, seed=seed, perm_seed=99).draw_dev(n)
        else:
            d_ids = torch.from_numpy(np.random.default_rng(seed).integers(0, nk, n, dtype=np.int32)).to(dev)
        self.seq = [(0, self.gen_now(s // GB, now0)) for s in range(total)]
        self.h_ids = d_ids.view(total, B).cpu().numpy()
        d_keys = torch.empty(n * L + 16, dtype=torch.uint8, device=dev)
        d_keys[-16:] = 0
        step = 256
        for lo in range(0, total, step):
            hi = min(total, lo + step)
            d_keys[lo * B * L:hi * B * L] = self.d_keytab.index_select(0, d_ids[lo * B:hi * B]).reshape(-1)
        self.d_keys = d_keys
        self.distinct_keys = int(torch.unique(d_ids).numel())
        samp = list(range(0, ngen, max(1, ngen // 16)))[:16]            # (a generation touches the tables once per distinct key: its shares have disjoint keys)
        self.access_frac = float(np.mean([int(torch.unique(d_ids[g * G:(g + 1) * G]).numel()) for g in samp])) / G
        del d_ids
        t_offG = torch.from_numpy((np.arange(G + 1, dtype=np.int64) * L).astype(np.int32)).to(dev)
        self.cols = {"off": t_offG.repeat(ngen), "hits": torch.full((n,), 1, dtype=torch.int64, device=dev),
                     "limit": torch.full((n,), 100, dtype=torch.int64, device=dev),
                     "duration": torch.full((n,), self.duration_ms, dtype=torch.int64, device=dev),
                     "algorithm": torch.full((n,), self.algo_id, dtype=torch.uint8, device=dev),
                     "behavior": torch.zeros((n,), dtype=torch.int32, device=dev)}
        c, ga = self.cols, self.ga
        self.gens = [ga.GuberBatch(G, 0, d_keys.data_ptr() + g * G * L, c["off"].data_ptr() + g * (G + 1) * 4, c["hits"].data_ptr() + g * G * 8,
                                   c["limit"].data_ptr() + g * G * 8, c["duration"].data_ptr() + g * G * 8, None, None, c["algorithm"].data_ptr() + g * G,
                                   c["behavior"].data_ptr() + g * G * 4, None, None, None, int(self.gen_now(g, now0))) for g in range(ngen)]
        # every generation writes its own answer arrays (request order); every timed slice of B is digest-checked afterwards
        self.res_all = {"status": torch.empty(n, dtype=torch.uint8, device=dev), "err": torch.empty(n, dtype=torch.uint8, device=dev),
                        "limit": torch.empty(n, dtype=torch.int64, device=dev), "remaining": torch.empty(n, dtype=torch.int64, device=dev),
                        "reset_time": torch.empty(n, dtype=torch.int64, device=dev)}
        r = self.res_all
        self.gres = [ga.GuberResult(r["status"].data_ptr() + g * G, r["limit"].data_ptr() + g * G * 8, r["remaining"].data_ptr() + g * G * 8,
                                    r["reset_time"].data_ptr() + g * G * 8, r["err"].data_ptr() + g * G, 0, 0, 0, 0, 0) for g in range(ngen)]
        torch.cuda.synchronize(dev)

    def keep_results(self, which):
        """the timed slices' answers ARE slices of the generations' answer arrays (request order)"""
        which = list(which)
        B = self.ctx.B
        lo = which[0]
        assert which == list(range(lo, lo + len(which)))
        self.res_cols = {k: v[lo * B:(lo + len(which)) * B] for k, v in self.res_all.items()}
        self.kept = {s: self.SliceResult(self, k) for k, s in enumerate(which)}

    def run(self, lo, hi, timed=False):
        torch, GB = self.torch, self.GB
        assert lo % GB == 0 and hi % GB == 0
        g0, g1 = lo // GB, hi // GB
        N = g1 - g0
        ba = (self.ga.GuberBatch * max(N, 1))(*self.gens[g0:g1])
        ra = (self.ga.GuberResult * max(N, 1))(*self.gres[g0:g1])
        if timed:
            torch.cuda.synchronize(self.ctx.dev)
            self.ctx.barrier()
        t0 = time.perf_counter()
        c0 = time.thread_time()
        done = self.front.eval_dev(ba, ra, N)
        t_enq = time.perf_counter()
        self.last_enqueue_busy_s = time.thread_time() - c0
        self.front.synchronize()
        torch.cuda.synchronize(self.ctx.dev)
        t1 = time.perf_counter()
        assert done == N, (done, N)
        if timed:
            self.ctx.barrier()
            self.last_stream_ms, self.last_stream_batches = [(t1 - t0) * 1e3], [hi - lo]
            self.last_enqueue_s = t_enq - t0
            return t1 - t0, (t1 - t0) * 1e3
        return t1 - t0, None

    def kernel_profile(self):
        out = super().kernel_profile()
        self.gen_us = self.front.latencies()
        return out

    def latency_under_load(self):
        lat = sorted(getattr(self, "gen_us", []))
        if not lat:
            return None
        return {"unit": "us", "p50": round(percentile(lat, 0.5), 2), "p99": round(percentile(lat, 0.99), 2), "min": round(lat[0], 2), "max": round(lat[-1], 2),
                "n": len(lat), "what": (f"per generation of {self.G} requests: first routing kernel's start -> the end of the answers' last hop (HIP events on the routing "
                                        "stream), the whole profile segment enqueued at once like the timed region: the routing runs two generations ahead of the "
                                        "evaluation, so a generation's way includes the two before it (open loop)")}

    def latency(self):
        """one generation at a time, nothing else in flight: the call -> the answers in HBM (host clock: enqueue included)"""
        GB = self.GB
        lo = (self.warmup + self.steps + self.profile_steps) // GB
        hi = lo + self.latency_steps // GB
        lat = []
        for g in range(lo, hi):
            ba, ra = (self.ga.GuberBatch * 1)(self.gens[g]), (self.ga.GuberResult * 1)(self.gres[g])
            t0 = time.perf_counter()
            self.front.eval_dev(ba, ra, 1)
            self.front.synchronize()
            lat.append((time.perf_counter() - t0) * 1e6)
        if not lat:
            return None
        lat.sort()
        return {"unit": "us", "p50": round(percentile(lat, 0.5), 2), "p99": round(percentile(lat, 0.99), 2), "min": round(lat[0], 2), "n": len(lat),
                "what": f"one generation of {self.G} requests (each a fresh part of the stream), host clock around guber_front_eval_dev + guber_front_synchronize, nothing else in flight"}

    def close(self):
        self.front_stats = self.front.stats()
        self.front.close()
        self.gens, self.gres, self.res_all, self.res_cols = [], [], None, None
        super().close()


def oracle_populate(rig, orc, threads, now0):
    ctx = rig.ctx
    for lo in range(0, len(ctx.my_ids), 1 << 18):
        orc.eval(rig.streams.bench_batch(ctx.table, ctx.my_ids[lo:lo + (1 << 18)], now0, hits=0, algorithm=rig.algo_id, duration=rig.duration_ms),
                 threads=threads)


def pipeline_traffic(tj, algo, kernels, launches, per_launch, B):
    """HBM-side bytes per step from the PMC passes (profiles/roofline_traffic.json, written by tools/summarize_r04.py from separate
    --pmc runs: bytes per 65536-request batch and kernel).  A batch goes through ONE pipeline — the two launches with claims or the
    three owner-partitioned ones — so the step's traffic is the mix the profile segment ran: every kernel's bytes per batch weighted
    by the batches it carried, over the batches that entered a pipeline (those of the first-stage kernels).  Fused launches use the
    per-batch figures measured on fused launches where the file has them ("<algo>_fused"), the one-table figures otherwise."""
    one, fused = tj.get(algo, {}), tj.get(algo + "_fused", {})
    total = entered = 0.0
    for k in kernels:
        base = k.replace("_multi", "")
        if base == "k_evalpart":          # GUBER_FUSE_EP: one batch's k_eval3 and the next one's k_part in one launch — priced as the two
            parts = [fused.get(b) or one.get(b) for b in ("k_eval3", "k_part")]
            per_batch = sum(parts) if all(parts) else None
        else:
            per_batch = (fused.get(base) if k.endswith("_multi") else None) or one.get(base)
        if not per_batch:
            return None
        batches = launches.get(k, 0) * per_launch.get(k, B) / B
        total += per_batch * batches
        if base in ("k_front", "k_part", "k_evalpart"):
            entered += batches
    return int(total / entered * B / 65536) if entered else None


def parity_over_timed_work(rig, orc, threads, now0, label):
    """The oracle is fed what the engine was fed — populate, warm-up, every timed batch, in order, with the same clocks — and
    the engine's kept answers (every 64th timed batch, the first 8, the last) must equal the oracle's element-wise.
    -> (ok, compared batches, oracle seconds for the timed batches)"""
    import support
    oracle_populate(rig, orc, threads, now0)
    got = rig.result_digests()                                       # device-side digest of EVERY timed batch's five result columns
    ok, compared, el = True, 0, 0.0
    for s in range(0, rig.warmup + rig.steps):
        hb = rig.host_batch(s)
        t0 = time.perf_counter()
        want = orc.eval(hb, threads=threads)
        if s >= rig.warmup:
            el += time.perf_counter() - t0
        if s in rig.kept:
            compared += 1
            if host_digest(want) != got[s]:                          # element-wise only to say where
                ok = False
                try:
                    support.assert_results_equal(rig.kept[s].host(), want, f"{label} batch {s}")
                    print(f"PARITY FAILURE: {label} batch {s}: digests differ, columns equal (digest bug)", file=sys.stderr)
                except AssertionError as ex:
                    print("PARITY FAILURE:", ex, file=sys.stderr)
    return ok, compared, el


def measure_ceilings(keys):
    """what THIS box sustains, measured before anything else is on the GPU (tools/random_access quick, about a second): independent random
    accesses of the engine's shape (128-byte bucket read, its 64-byte record written back) over a table of the tables' size, and a streaming copy"""
    import subprocess
    exe = os.path.join(ROOT, "tools", "random_access")
    if not os.path.exists(exe):
        return None
    try:
        gb = max(0.25, 4.5 * keys / 10_000_000)                      # 2^25 slots x 144 B for 10 M keys
        p = subprocess.run([exe, "quick", f"{gb:.2f}"], capture_output=True, text=True, timeout=120)
        return json.loads(p.stdout.strip().splitlines()[-1])
    except Exception:   # noqa: BLE001
        return None


def against_ceilings(ceil, value, access_frac, bytes_per_decision):
    """a rate next to the box's measured ceilings: the table accesses it implies (one per distinct key of a batch) over the measured rate of
    independent random bucket accesses, and its algorithmic bytes over the measured streaming copy"""
    if not ceil or "error" in ceil or not value:
        return None
    acc = value * access_frac
    return {"table_accesses_per_decision": round(access_frac, 4), "table_accesses_per_s": round(acc, 1),
            "random_r128_w64_ceiling_per_s": round(ceil["random_r128_w64_Gbuckets_s"] * 1e9, 1),
            "frac_of_measured_random_rw": round(acc / (ceil["random_r128_w64_Gbuckets_s"] * 1e9), 4),
            "stream_copy_ceiling_GBps": ceil["stream_copy_GBps"],
            "frac_of_measured_stream": round(value * bytes_per_decision / 1e9 / ceil["stream_copy_GBps"], 4),
            "table_gb": ceil["table_gb"],
            "what": ("measured on this box at the start of this run (tools/random_access quick): INDEPENDENT random accesses of a 128-byte bucket read + its 64-byte record "
                     "written back over a table of the tables' size — what the table phase could do if nothing depended on anything — and a streaming copy (read + write) in "
                     "place of the 8 TB/s spec figure")}


def usable_cpus():
    """CPUs this process may really use: the affinity mask, capped by a cgroup CPU quota (cpu.max)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, -(-int(q) // int(per))))
    except Exception:   # noqa: BLE001
        pass
    return n


def cpu_baseline_sample(rig, orc, w, now0, seconds):
    """one configuration of the CPU baseline: a fresh oracle with W worker caches served by min(W, usable CPUs) threads (a thread
    owns the workers w mod T, as goroutines share cores), the same resident keys, the timed stream's batches from its
    beginning for a bounded time"""
    th = min(w, usable_cpus()) if w > 1 else 0
    oracle_populate(rig, orc, th, now0)
    done, el, t_all = 0, 0.0, time.perf_counter()
    for s in range(rig.warmup, rig.warmup + rig.steps):
        hb = rig.host_batch(s)                                       # (building the host arrays is not the baseline's work)
        t0 = time.perf_counter()
        orc.eval(hb, threads=th)
        el += time.perf_counter() - t0
        done += 1
        if el >= seconds or time.perf_counter() - t_all >= 4 * seconds:
            break
    return done * rig.ctx.B / el, done, el


def finish_distributed(dist):
    """The line is out: leave without the interpreter's teardown.  With several processes on a node the order in which the process
    group, the HIP runtime and the library's statics go away at exit is not ours to choose (an abort there — seen once with two
    ranks sharing one GPU — would turn a finished measurement into a failed run)."""
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:   # noqa: BLE001
        pass
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0)


def rocprof_reference(kernel):
    """the committed rocprofv3 summary of this command (profiles/r06_rocprof_summary.json), if any: the average launch duration of `kernel`
    (the one the line's `roofline.kernel` names) from the trace, beside the kernel the trace's GPU spent most time in — so that the line and
    the file can be checked against each other.  (The traced run is slower than an untraced one — the profiler's interception costs the host
    that enqueues ~25 launches per generation —, so its kernels see less contention: its averages are a few microseconds below the line's.)"""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "r06_rocprof_summary.json")))
        ks = (j.get("kernels") or {}).get("routed") or {}
        k = ks.get(kernel)
        dom = j.get("dominant_kernel") or {}
        return {"file": "profiles/r06_rocprof_summary.json", "kernel": kernel,
                "avg_us": round(k["avg_us"], 3) if k else None, "max_us": round(k["max_us"], 3) if k else None, "launches": k["launches"] if k else None,
                "trace_dominant_kernel": {"name": dom.get("name"), "avg_us": dom.get("avg_us"), "max_us": dom.get("max_us"), "launches": dom.get("launches")},
                "command": j.get("command"), "bench_line_of_traced_run": (j.get("bench_lines") or {}).get("routed")}
    except Exception:   # noqa: BLE001
        return None


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
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # invoked plainly (`python bench.py --gpus N`, the shape of the driver's 1-GPU command): become the launcher the contract
        # names — one rank per GPU of this node under torch.distributed.run, rendezvous on 127.0.0.1
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} (or plainly: bench.py starts it)")
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
    ctx.router = args.router
    ctx.barrier = (lambda: dist.barrier()) if world > 1 else (lambda: None)
    ctx.max_over_ranks = lambda v: shard.max_over_ranks(v, device=red_dev)
    K, B = args.keys, args.batch
    GSYNC = args.global_sync
    NOW0 = streams.NOW0

    def route_on_device(eng, ring, kb, ko):       # ReplicatedConsistentHash.Get for a chunk of keys (k_route)
        d_kb, d_ko = torch.from_numpy(kb).to(dev), torch.from_numpy(ko.view(np.int32)).to(dev)
        d_owner = torch.empty(len(ko) - 1, dtype=torch.int32, device=dev)
        torch.cuda.current_stream(dev).synchronize()                  # (the copies are torch's stream's, the router kernel the engine's)
        eng.route_dev(ring, d_kb.data_ptr(), d_ko.data_ptr(), len(ko) - 1, d_owner.data_ptr())
        return d_owner.cpu().numpy()
    ctx.route_on_device = route_on_device

    # ---- key ownership: ids of the global key space (world x K) this rank owns on the ring ------
    peers = max(world, args.ring_peers or world)
    total_keys = K if GSYNC else K * peers      # GLOBAL: one key space, replicated on every GPU
    ctx.table = streams.key_table(total_keys)
    if GSYNC or peers == 1:
        ctx.my_ids = np.arange(total_keys, dtype=np.int64)
    else:
        tmp = ga.Engine(cache_size=1024, device=local_rank, max_batch=1024)
        ctx.my_ids = shard.owned_key_ids(ctx.table, peers, rank, route=lambda ring, kb, ko: route_on_device(tmp, ring, kb, ko), chunk=4_000_000)
        tmp.close()

    if GSYNC:
        return run_global(args, ctx, dist)

    S = max(1, args.shards)
    seed = 1234 + rank * 64
    ctx.ceil = measure_ceilings(K) if (rank == 0 and world == 1) else None
    routed = args.headline == "routed"
    GB = max(1, args.gen_batches) if routed else 1
    up = lambda v: -(-int(v) // GB) * GB                              # noqa: E731  (whole generations)
    steps = up(max(args.steps, args.min_batches))
    rig = RoutedRig(ctx, args.algo, args.dist, S, GB, duration_ms=args.duration_ms) if routed else Rig(ctx, args.algo, args.dist, S, duration_ms=args.duration_ms)
    resident = rig.populate(NOW0)
    if os.environ.get("GUBER_BENCH_EXIT_AFTER_SETUP"):              # scripts/gpu_r06_fault.sh: many fresh starts of the set-up phase
        torch.cuda.synchronize(dev)
        print(json.dumps({"setup_only": True, "resident": int(resident)}), flush=True)
        os._exit(0)
    # what every rank holds, and how many ranks the collective backend (nccl = RCCL for N > 1) really sees
    resident_by_rank = shard.gather_over_ranks(resident, device=red_dev)
    ranks_seen = shard.sum_over_ranks(1, device=red_dev)
    # --warmup W is honoured as a LOWER bound: the driver's command passes 5, and five batches do not reach every one of the S shards
    # (nor every stream with every kernel) before the clock starts — the first process on a fresh box then pays first-use costs inside
    # the timed region (measured: 9.4-9.6 instead of 10.0-10.2 G/s, profiles/r05_t_*).  At least four batches per shard run untimed.
    warm = up(max(args.warmup, 4 * S))
    m = rig.measure(steps, warm, NOW0, seed, profile_steps=up(max(0, args.profile_steps)), latency_steps=up(max(0, args.latency_steps)))

    if os.environ.get("GUBER_BENCH_EXIT_AFTER_TIMED"):              # scripts/gpu_r06_fault.sh: many fresh starts up to the end of the timed region
        print(json.dumps({"timed_only": True, "value": m["value"]}), flush=True)
        os._exit(0)
    roofline = latency = cpu = parity = None
    extras = {}
    if rank == 0:
        # ---- per-kernel durations: HIP events around every launch of the profile segment, dispatched like the timed region ----
        fused = args.dispatch == "one" and S > 1
        kernel_ms, per_launch, launches = rig.kernel_profile()
        latency = rig.latency()
        cand = {k: v for k, v in kernel_ms.items() if k in KERNEL_BYTES[args.algo] and v > 0}
        if cand:
            # the dominant kernel = the one the GPU spends most time in; its bytes per launch = bytes per request x the
            # requests one launch carries (a fused launch carries the batches of up to four shards)
            dom = max(cand, key=lambda k: cand[k] * launches.get(k, 1))
            dom_bytes = int(KERNEL_BYTES[args.algo][dom] * per_launch.get(dom, B))
            achieved = dom_bytes / (cand[dom] * 1e-3) / 1e9
            traffic = measured = None
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "roofline_traffic.json")))
                if routed:                                           # every kernel of the routed pipeline incl. the front's copies, per 65536-request batch
                    traffic = int(tj["routed"]["corrected_per_batch"] * B / 65536) if args.algo == "token" else None
                else:
                    traffic = pipeline_traffic(tj, args.algo, cand, launches, per_launch, B)
                measured = tj.get("note")
            except Exception:   # noqa: BLE001
                pass
            # the issue side: what the SIMDs spend issuing the three kernels' instructions for one batch (SQ counter passes, committed as
            # profiles/roofline_issue.json by tools/summarize_r05.py) over the step the driver's clock sees
            issue = None
            try:
                ij = json.load(open(os.path.join(ROOT, "profiles", "roofline_issue.json")))
                if ij.get("arrangement", "presplit") != args.headline:
                    raise KeyError("the committed SQ passes are of the other arrangement")
                iu = float(ij["issue_us_per_batch"]) * B / 65536
                issue = {"issue_us_per_step": round(iu, 4), "step_us": round(m["ms_per_step"] * 1e3, 4), "frac": round(iu / (m["ms_per_step"] * 1e3), 4),
                         "insts_per_wave": {k: v["insts_per_wave_all"] for k, v in ij["kernels"].items()},
                         "what": ("SQ_ACTIVE_INST_ANY of every kernel of the pipeline (the front's copies included) per 65536-request batch (quad-cycles a SIMD spent issuing, summed over the chip) x 4 cycles / "
                                  "(1024 SIMDs x 2.4 GHz) over the step time: the share of the step in which EVERY SIMD of the chip would have to be issuing — the resource "
                                  "this pipeline is closest to filling (round 5: fewer instructions per request moved the rate almost one for one, DESIGN.md section 4)"),
                         "source": ij.get("source"), "counters_from": "a separate rocprofv3 --pmc run of this command (committed), not this run"}
            except Exception:   # noqa: BLE001
                pass
            pipe = BYTES_PER_DECISION[args.algo] * B / (m["ms_per_step"] * 1e-3) / 1e9
            # the line's roofline: the WHOLE pipeline's algorithmic bytes over the driver-visible step time — reproducible from
            # `ms_per_step` alone.  The per-kernel figure of a fused run is a diagnostic: the streams' kernels overlap in time.
            roofline = {"bound": "hbm", "achieved": round(pipe, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(pipe / HBM_PEAK_GBPS, 6),
                        "what": (f"{BYTES_PER_DECISION[args.algo]} algorithmic B per decision (SURVEY 8d) x {B} decisions per step / ms_per_step: every kernel of "
                                 "the pipeline, all shards overlapping, as the driver's clock sees it"),
                        "measured": against_ceilings(ctx.ceil, m["value"], rig.access_frac, BYTES_PER_DECISION[args.algo]),
                        "traffic": traffic, "traffic_note": measured,
                        "issue": issue,
                        "limiter": (("the dependencies between the four in-order queues (the HIP runtime gives a process four hardware queues: one routing stream + three for the "
                                     "tables): every queue is busy ~60 % of the stretch, 2.3 kernels run at a time (profiles/r06_front_timeline.txt); the routed pipeline does the "
                                     "pre-split one's work plus the front's copies (`traffic` counts them), and the pre-split one is bound by instruction issue (`presplit`, DESIGN.md section 4)")
                                    if routed else
                                    ("instruction issue, not HBM bytes: the pipeline without any table access is not faster and fewer fabric transactions barely moved it "
                                     "(round 4: profiles/r04_w_*, r04_x_*), while round 5's instruction diet — 15 % fewer instructions per request in the three kernels — "
                                     "gave 12.5 % on one box (profiles/r05_d_*, r05_e_*); `issue.frac` is how full the SIMDs' issue slots are on average, the rest is "
                                     "imbalance between owners, kernel tails on three streams and waves parked at s_waitcnt")),
                        "kernel": dom,
                        "dominant_kernel_overlapped": {
                            "kernel": dom, "achieved": round(achieved, 2), "frac": round(achieved / HBM_PEAK_GBPS, 6),
                            "algorithmic_bytes_per_launch": dom_bytes, "bytes_per_request": KERNEL_BYTES[args.algo][dom],
                            "requests_per_launch": round(per_launch.get(dom, B), 1),
                            "note": ("DIAGNOSTIC: launch durations of a run whose streams overlap include the other streams' contention and cannot be summed "
                                     "to the step time; the exclusive figures are `shards_1.roofline_frac` (one table, one batch in flight)") if fused else
                                    "one table, one batch in flight: exclusive kernel time"},
                        "kernel_avg_us": {k: round(v * 1e3, 2) for k, v in kernel_ms.items() if v > 0},
                        "launches_profiled": launches,
                        "kernel_timing": (f"HIP events around every launch of {args.profile_steps} further distinct batches dispatched exactly like the timed "
                                          "region (all shards' streams overlapping; a launch carries the next batch of up to four shards)") if fused
                        else f"HIP events around every launch of {args.profile_steps} further distinct batches on the engine stream, one batch in flight",
                        "rocprof": rocprof_reference(dom) if routed else None}
        if latency is not None:
            latency = {"idle": latency, "under_load": rig.latency_under_load(),
                       "note": "the BASELINE metric pairs decisions/s with p99 batch latency: `under_load` is the latency in the regime `value` is measured in"}

    # ---- parity over the timed work on EVERY rank (ranks own disjoint keys, replicated_hash.go:104-119: each checks its own timed
    # stream against its own oracle) + the CPU baseline (rank 0; the W sweep at N = 1 only) ---------------------------
    parity_by_rank = None
    if world > 1 and not args.no_cpu_baseline:
        import support
        gate_w = min(os.cpu_count() or 1, 32)
        th = max(1, min(gate_w, usable_cpus() // world))                     # (the ranks run their passes at the same time on one host)
        orc = support.Oracle(cache_size=4 * K, workers=gate_w)
        ok, compared, el = parity_over_timed_work(rig, orc, th if gate_w > 1 else 0, NOW0, f"rank {rank}")
        orc.close()
        ok = bool(ok and m["internal_retries"] == 0 and compared == steps)
        oks = shard.gather_over_ranks(1 if ok else 0, device=red_dev)
        cmp_by_rank = shard.gather_over_ranks(compared, device=red_dev)
        if not all(oks):
            raise SystemExit(f"parity gate failed on rank(s) {[r for r, o in enumerate(oks) if not o]}: refusing to report a number")
        parity_by_rank = cmp_by_rank
        parity = (f"{sum(oks)}/{world} ranks, {min(cmp_by_rank)}/{steps} timed batches each: bit-exact vs the rank's own oracle fed the rank's whole stream "
                  f"(populate, warm-up, all timed batches in order) by a 64-bit digest of status|err, limit, remaining, reset_time computed on the device and on "
                  f"the oracle's answers; internal retries in the timed region: 0 on every rank")
        if rank == 0:
            ucpu = usable_cpus()
            cpu = {"value": round(steps * B / el, 1), "unit": "decisions/s", "cores": th, "kind": "port", "workers": gate_w,
                   "host": {"cpus": os.cpu_count(), "usable": ucpu},
                   "sample": f"rank 0's parity pass: all {steps} timed batches of {B} ({el:.1f} s), {K} resident keys, W = {gate_w} worker caches on {th} threads "
                             f"while the other {world - 1} rank(s) ran theirs on the same host (the N = 1 line carries the thread sweep)"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import support
        ncpu = os.cpu_count() or 1
        ucpu = usable_cpus()
        ws = []
        for tok in args.cpu_threads.split(","):
            tok = tok.strip()
            if tok:
                w = ncpu if tok == "all" else max(1, min(ncpu, int(tok)))
                if w not in ws:
                    ws.append(w)
        gate_w = min(ncpu, 32)
        orc = support.Oracle(cache_size=4 * K, workers=gate_w)
        ok, compared, el = parity_over_timed_work(rig, orc, min(gate_w, ucpu) if gate_w > 1 else 0, NOW0, "headline")
        orc.close()
        ok = ok and m["internal_retries"] == 0
        parity = (f"bit-exact vs the oracle fed the whole stream (populate, warm-up, all {steps} timed batches in order): {compared}/{steps} timed batches by a "
                  f"64-bit digest of status|err, limit, remaining, reset_time computed on the device and on the oracle's answers (element-wise only on a "
                  f"mismatch), internal retries in the timed region: {m['internal_retries']}") if ok else "FAILED"
        if not ok:
            raise SystemExit("parity gate failed: refusing to report a number")
        res = {gate_w: (steps * B / el, steps, el)}
        for w in ws:
            if w == gate_w:
                continue
            o = support.Oracle(cache_size=4 * K, workers=w)
            res[w] = cpu_baseline_sample(rig, o, w, NOW0, args.cpu_seconds)
            o.close()
        best = max(res, key=lambda w: res[w][0])
        cpu = {"value": round(res[best][0], 1), "unit": "decisions/s", "cores": min(best, ucpu), "kind": "port", "workers": best,
               "host": {"cpus": ncpu, "usable": ucpu, "note": "usable = affinity mask capped by the cgroup CPU quota (cpu.max); W workers are served by min(W, usable) threads"},
               "sample": f"{res[best][1]} batches of {B} from the beginning of the timed stream ({res[best][2]:.1f} s), {K} resident keys, the oracle in the "
                         f"reference's worker-sharded design (W caches, XXH64-range sharding, workers.go:180-184; persistent threads, a parallel stable partition "
                         f"per batch); W = {gate_w} ran the whole timed stream (it is the parity pass); host has {ncpu} cores of which {ucpu} are usable",
               "by_threads": {str(w): {"value": round(v[0], 1), "batches": v[1], "seconds": round(v[2], 2)} for w, v in sorted(res.items())}}

    touched = m["distinct_keys_in_stream"]
    if routed:
        wl = (f"{K} resident keys per GPU, ONE {args.dist} request stream" + (" s=1.1" if args.dist == "zipf" else "") + f" over them in ARRIVAL order, never split on the host, "
              f"batch={B}, {args.algo.upper()}_BUCKET, hits=1 limit=100 duration={args.duration_ms}ms, {world}xMI355X"
              + (f", keys sharded by replicated consistent hash (512 vnodes, fnv1, {peers} peers)" if peers > 1 else "")
              + f"; generations of {GB} batches (what a batcher has collected while the previous generation ran, peer_client.go:284-337; one now_ms each) go through "
                f"guber_front_eval_dev, everything inside the clock: XXH64 of every HashKey + the placement's rule -> one of {S} logical shards (workers.go:261-289, getWorker "
                f":180-184), shares contiguous and in arrival order, the {S} tables through the fused launches on {args.streams} streams, answers back in REQUEST order "
                f"(gubernator.proto:51-54)")
    else:
        wl = (f"{K} resident keys per GPU, one {args.dist} request stream" + (" s=1.1" if args.dist == "zipf" else "") +
              f" over them, batch={B}, {args.algo.upper()}_BUCKET, hits=1 limit=100 duration={args.duration_ms}ms, {world}xMI355X"
              + (", keys sharded by replicated consistent hash (512 vnodes, fnv1)" if world > 1 else "")
              + ((f", {S} logical shards per GPU (own table each; the stream is split key by key by the placement OUTSIDE the clock, a shard flushes a batch "
                  f"when {B} requests are waiting; answers stay in the shards' order), " +
                  ("own stream + batcher thread each" if args.dispatch == "threads" else
                   f"one dispatcher, shards spread over {args.streams} stream(s): the next batch of up to four shards of a stream per pair of launches")) if S > 1 else ", one table"))
    headline_cfg = {"workload": wl, "arrangement": args.headline, "generation_requests": (GB * B) if routed else None,
                    "keys_per_gpu": K, "ring_peers": peers, "batch": B, "algorithm": args.algo, "resident_items_rank0": int(resident),
                    "resident_items_by_rank": resident_by_rank, "ranks_seen_by_the_collective_backend": ranks_seen, "backend": args.backend if world > 1 else None,
                    "logical_shards_per_gpu": S, "dispatch": args.dispatch if S > 1 else "caller thread", "placement": rig.placement, "host_cores": os.cpu_count(),
                    # engine options taken from the environment (experiments; all unset in the driver's run)
                    "engine_env": {k: os.environ[k] for k in ("GUBER_FUSE_EP", "GUBER_PIPELINE", "GUBER_PT_BITS", "GUBER_HIP_LIB") if k in os.environ},
                    "stream": {"replayed": False, "distinct_batches_total": len(rig.seq), "timed_batches": steps,
                               "distinct_keys_touched": touched, "table_bytes_touched": touched * 144, "table_bytes_touched_in_64B_sectors": touched * 192,
                               "now_ms": "advances 1 ms per batch" if not routed else f"one per generation, advancing {GB} ms per generation",
                               "routing": ("per request ON THE DEVICE, inside the clock (guber_front_*): hash, placement, shares, answers back in request order" if routed else
                                           "per-request routing to the shards is outside the clock (the front end's work: see `pool`); `shards_1` needs none")}}
    front_stats = None
    rig.close()
    if routed:
        front_stats = rig.front_stats
    del rig
    torch.cuda.empty_cache()

    # ---- extras (rank 0 of a 1-GPU run): other configurations, same machinery ----
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
            "value": round(m["value"], 1), "unit": "decisions/s", "n_gpus": world, "steps": steps, "steps_requested": args.steps,
            "warmup": warm, "warmup_requested": args.warmup, "ms_per_step": round(m["ms_per_step"], 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "int64" if args.algo == "token" else "f64", "data": "synthetic",
            "config": headline_cfg,
            "timed_region": {"distinct_batches": steps, "replays": 0, "ms": round(m["timed_ms"], 3), "untimed_batches_before": warm,
                             "untimed_note": (f"--warmup {args.warmup} was raised to {warm} untimed batches (four per shard: every shard, stream and kernel has run before the clock starts)"
                                              if warm != args.warmup else "the untimed batches are the --warmup asked for"),
                             "note": (f"--steps {args.steps} was raised to {steps}: the timed region is never shorter than {args.min_batches} distinct batches"
                                      if steps != args.steps else "every timed batch is a distinct part of the stream"),
                             "ms_per_step_hip_events": round(m["ms_per_step_events"], 5), "host_enqueue_ms": round(m["enqueue_ms"], 3), "host_enqueue_busy_ms": round(m["enqueue_busy_ms"], 3),
                             "host_enqueue_note": "wall time of the ONE dispatcher call that enqueues the whole timed region, and the CPU time its thread burnt in it (busy << wall: it waits for queue room, the GPU is the bound; busy ~ wall: the host is)",
                             "front": front_stats,
                             "enqueue": ("ONE guber_front_eval_dev call for the whole timed region: routing launches ahead of the evaluation, the shares' sizes read from pinned memory, "
                                         f"one dispatcher for {S} shards over {args.streams} stream(s)") if routed else (
                                         "caller thread" if S == 1 else f"{S} pre-started batcher threads behind a barrier" if args.dispatch == "threads"
                                         else f"one dispatcher for {S} shards over {args.streams} stream(s) (guber_eval_batches_routed_dev: batches of shards that share a stream share launches)"),
                             "shard_streams": m["shard_streams"]},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "parity_batches_by_rank": parity_by_rank, "batch_latency": latency,
        }
        if world > 1 and parity is None and not args.no_cpu_baseline:
            raise SystemExit("no parity verdict for an N > 1 run: refusing to report a number")
        out.update(extras)
        if routed and isinstance(extras.get("presplit"), dict) and extras["presplit"].get("value"):
            # what the per-request routing on the device and the answers' way back into request order cost
            out["routed_over_presplit"] = round(out["value"] / extras["presplit"]["value"], 4)
        print(json.dumps(out), flush=True)
    if world > 1:
        finish_distributed(dist)


def run_extra(name, args, ctx, NOW0, seed):
    """one extra configuration -> sub-object of the JSON line"""
    import support
    K, B = ctx.K, ctx.B
    steps = max(64, args.extra_batches)
    warmup = max(4, min(args.warmup, 16))                        # (raised to four batches per shard below)
    if name == "routed":
        return run_routed(args, ctx, NOW0, seed, steps)
    if name == "routed_leaky":                                      # BASELINE configs[2] in the all-inclusive arrangement
        return run_routed(args, ctx, NOW0, seed, steps, algo="leaky", dist_kind="zipf")
    if name in ("global_sync", "two_ranks"):
        return run_rehearsal(name, args)
    if name == "end_to_end":
        return run_end_to_end(args, ctx, NOW0, seed)
    if name == "pool":
        return run_pool(args)
    algo, dist_kind, S, dur = {"leaky": ("leaky", "zipf", max(1, args.shards), 60_000), "shards_1": (args.algo, args.dist, 1, args.duration_ms),
                               "presplit": (args.algo, args.dist, max(1, args.shards), args.duration_ms),
                               "uniform": (args.algo, "uniform", max(1, args.shards), args.duration_ms),
                               "expiring": (args.algo, "zipf", max(1, args.shards), 500)}[name]
    rig = Rig(ctx, algo, dist_kind, S, duration_ms=dur)
    rig.populate(NOW0)
    prof = 64 if name in ("leaky", "shards_1", "presplit") else 0
    lat = 64 if name in ("leaky", "shards_1", "presplit") else 0
    warmup = max(warmup, 4 * S)
    m = rig.measure(steps, warmup, NOW0, seed, profile_steps=prof, latency_steps=lat)
    out = {"value": round(m["value"], 1), "unit": "decisions/s", "ms_per_step": round(m["ms_per_step"], 5), "timed_batches": steps, "replays": 0,
           "distinct_keys_touched": m["distinct_keys_in_stream"], "dtype": "int64" if algo == "token" else "f64",
           "workload": f"{K} keys, {dist_kind}, {algo.upper()}_BUCKET, batch {B}, duration {dur} ms, {S} logical shard(s), {steps} distinct batches"}
    if prof:
        km, per_launch, _ = rig.kernel_profile()
        out["kernel_avg_us"] = {k: round(v * 1e3, 2) for k, v in km.items() if v > 0}
        kb = KERNEL_BYTES[algo]
        out["roofline_frac"] = {k: round(kb[k] * per_launch.get(k, B) / (v * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5) for k, v in km.items() if v > 0 and k in kb}
        out["batch_latency"] = {"idle": rig.latency(), "under_load": rig.latency_under_load()}
    out["measured"] = against_ceilings(getattr(ctx, "ceil", None), m["value"], rig.access_frac, BYTES_PER_DECISION[algo])
    if name == "presplit":
        out["excludes"] = ("the per-request routing to the shards (the stream is split by the placement before the clock starts) and the answers' way back into request "
                           "order (they stay in the shards' order): rounds 2-5's headline, kept as the kernel pipelines' own rate; the line's `value` includes both")
    if not args.no_cpu_baseline:                                    # every printed number is gated on ITS OWN timed work
        w = min(os.cpu_count() or 1, 32)
        orc = support.Oracle(cache_size=4 * K, workers=w)
        ok, compared, _ = parity_over_timed_work(rig, orc, min(w, usable_cpus()) if w > 1 else 0, NOW0, name)
        orc.close()
        ok = ok and m["internal_retries"] == 0
        out["parity"] = (f"bit-exact vs the oracle fed the whole stream: {compared}/{steps} timed batches by digest (tolerance 0), internal retries {m['internal_retries']}"
                         if ok else "FAILED")
        if name == "expiring":
            out["renewals"] = "duration 500 ms, now_ms +1 per batch: every touched bucket expires and is recreated about every 500 batches under the clock"
        if not ok:
            out.pop("value")
    rig.close()
    return out


def run_routed(args, ctx, NOW0, seed, steps, algo=None, dist_kind=None):
    """the all-inclusive arrangement (RoutedRig): one raw stream -> device route -> S tables -> answers in request order, all inside the
    clock; parity against the oracle fed the same stream in request order, every timed batch by digest"""
    import support
    K, B = ctx.K, ctx.B
    algo, dist_kind = algo or args.algo, dist_kind or args.dist
    S, GB = max(1, args.shards), max(1, args.gen_batches)
    steps = -(-steps // GB) * GB
    rig = RoutedRig(ctx, algo, dist_kind, S, GB, duration_ms=args.duration_ms)
    rig.populate(NOW0)
    m = rig.measure(steps, 4 * GB, NOW0, seed, profile_steps=32 * GB, latency_steps=16 * GB)
    pipe = BYTES_PER_DECISION[algo] * B / (m["ms_per_step"] * 1e-3) / 1e9
    out = {"value": round(m["value"], 1), "unit": "decisions/s", "ms_per_step": round(m["ms_per_step"], 5), "timed_batches": steps, "replays": 0,
           "generation_requests": rig.G, "generations_timed": steps // GB, "distinct_keys_touched": m["distinct_keys_in_stream"],
           "dtype": "int64" if algo == "token" else "f64",
           "roofline": {"bound": "hbm", "achieved": round(pipe, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(pipe / HBM_PEAK_GBPS, 6),
                        "what": f"{BYTES_PER_DECISION[algo]} algorithmic B per decision x {B} / ms_per_step; the routing's copies (request columns into the shares, answers "
                                "back into arrival order) are coordination, not algorithmic bytes: they show in the time, not in the numerator"},
           "host_enqueue_ms": round(m["enqueue_ms"], 3), "host_enqueue_busy_ms": round(m["enqueue_busy_ms"], 3), "timed_ms": round(m["timed_ms"], 3),
           "workload": (f"{K} keys, ONE {dist_kind} request stream in arrival order, never split on the host, {algo.upper()}_BUCKET, generations of {GB} x {B} requests "
                        f"(one now_ms each, +{GB} ms per generation) -> guber_front_eval_dev: k_fr_count (XXH64 of every HashKey + the placement's rule -> shard, "
                        f"workers.go:180-184) + k_fr_scatter (shares contiguous, arrival order kept) on a routing stream two generations ahead -> the {S} tables "
                        f"through the fused launches on {args.streams} streams -> k_fr_out (answers in request order, gubernator.proto:51-54); {steps} distinct batches "
                        f"= {steps // GB} generations inside the clock, one call"),
           "placement": rig.placement}
    km, per_launch, launches = rig.kernel_profile()
    out["kernel_avg_us"] = {k: round(v * 1e3, 2) for k, v in km.items() if v > 0}
    out["requests_per_launch"] = {k: round(per_launch.get(k, 0), 1) for k, v in km.items() if v > 0}
    out["batch_latency"] = {"idle": rig.latency(), "under_load": rig.latency_under_load()}
    if not args.no_cpu_baseline:
        w = min(os.cpu_count() or 1, 32)
        orc = support.Oracle(cache_size=4 * K, workers=w)
        ok, compared, _ = parity_over_timed_work(rig, orc, min(w, usable_cpus()) if w > 1 else 0, NOW0, "routed")
        orc.close()
        ok = ok and m["internal_retries"] == 0 and compared == steps
        out["parity"] = (f"bit-exact vs ONE oracle fed the whole stream in request order (populate, warm-up, every timed generation): {compared}/{steps} timed batches "
                         f"by digest of the answers in REQUEST order (tolerance 0), internal retries {m['internal_retries']}" if ok else "FAILED")
        if not ok:
            out.pop("value")
    rig.close()
    out["front"] = rig.front_stats
    return out


def run_rehearsal(name, args):
    """N > 1 readiness on a box with ONE GPU (no multi-GPU hardware has run this code): further bench.py processes of
    their own, parity-gated like any run.
      global_sync: BASELINE config 5 — every request GLOBAL, two logical ranks on this GPU, guber_global_sync every 8 batches on the native
                   exchange (device copies between the ranks: RCCL refuses two ranks on one device), replicas converged or no number
      two_ranks:   BASELINE config 4's shape — a ring of 8 peers over 8 x keys, TWO of them as processes sharing this GPU (gloo for the
                   timing collectives), each owning what the ring gives gpu<rank> (~ keys each), each gated against its own oracle
    Real RCCL between GPUs has never executed; what these legs pin is that the N > 1 code paths run and answer like the reference."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GUBER_BENCH_EXIT_AFTER_SETUP", "GUBER_BENCH_EXIT_AFTER_TIMED")}
    if name == "global_sync":
        cmd = ["--gpus", "1", "--global-sync", "8", "--logical-ranks", "2", "--keys", str(args.keys), "--steps", "64", "--warmup", "8"]
    else:
        cmd = ["--gpus", "2", "--one-device", "--backend", "gloo", "--ring-peers", "8", "--keys", str(args.keys), "--steps", "256", "--min-batches", "256", "--warmup", "8",
               "--extras", "", "--profile-steps", "0", "--latency-steps", "0", "--cpu-threads", "16", "--cpu-seconds", "1"]
    p = subprocess.run([sys.executable, os.path.abspath(__file__)] + cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or len(lines) != 1:
        return {"error": (p.stdout + p.stderr)[-600:], "command": "bench.py " + " ".join(cmd)}
    j = json.loads(lines[0])
    out = {"value": j["value"], "unit": "decisions/s", "ms_per_step": j["ms_per_step"], "n_gpus": j["n_gpus"], "parity": j["parity"], "command": "bench.py " + " ".join(cmd),
           "hardware": "ONE GPU: " + ("two logical ranks of one process, device copies between them" if name == "global_sync" else
                                      "two processes (two of the ring's eight peers) time-slicing it; gloo carries the timing collectives") +
                       " — a rehearsal of the N > 1 code paths, not a scaling number; RCCL between GPUs has not executed on any hardware yet"}
    if name == "global_sync":
        out["global_sync"] = j.get("global_sync")
    else:
        c = j["config"]
        out.update({"resident_items_by_rank": c.get("resident_items_by_rank"), "ranks_seen_by_the_collective_backend": c.get("ranks_seen_by_the_collective_backend"),
                    "ring_peers": c.get("ring_peers"), "parity_batches_by_rank": j.get("parity_batches_by_rank")})
    return out


def run_pool(args):
    """the drop-in surface: caller threads -> V1Instance::GetRateLimits -> GPUWorkerPool (C++), tools/bench_pool.cpp"""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "bench_pool_c")
    if not os.path.exists(exe):
        return {"error": "tools/bench_pool_c is not built (make -C gubernator_amd/csrc bench_pool)"}
    out = {}
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else round(int(q) / int(per), 2)
    except Exception:   # noqa: BLE001
        pass
    cases = {"rpc_1000": (64, 8, 1000, args.keys, 2.0), "rpc_1000_256_callers": (256, 8, 1000, args.keys, 2.0), "rpc_1000_12_shards": (64, 12, 1000, args.keys, 2.0),
             "rpc_1000_one_table": (64, 1, 1000, args.keys, 2.0), "rpc_1000_one_table_128_callers": (128, 1, 1000, args.keys, 2.0),
             "rpc_1": (16, 8, 1, args.keys, 1.0), "rpc_1_one_caller": (1, 8, 1, args.keys, 1.0),
             # the payload stage (guber_wire_pool_*): the callers hand over SERIALIZED GetRateLimitsReq messages and get serialized responses back
             "wire_rpc_1000": (64, 8, 1000, args.keys, 2.0, "wire"), "wire_rpc_1000_128_callers": (128, 8, 1000, args.keys, 2.0, "wire"),
             "wire_rpc_1000_192_callers": (192, 8, 1000, args.keys, 2.0, "wire"),
             "wire_rpc_1000_256_callers": (256, 8, 1000, args.keys, 2.0, "wire"), "wire_rpc_1000_one_table_256_callers": (256, 1, 1000, args.keys, 2.0, "wire"),
             "wire_rpc_1_one_caller": (1, 8, 1, args.keys, 1.0, "wire")}
    for label, case in cases.items():
        T, S, items, keys, secs = case[:5]
        api = case[5] if len(case) > 5 else "c"
        p = subprocess.run([exe, str(T), str(S), str(items), str(keys), str(secs), "200", api], capture_output=True, text=True, timeout=300)
        mm = re.search(r"([0-9.]+) M decisions/s,\s+([0-9.]+) batches/s, avg batch\s+([0-9.]+) requests, errors (\d+)(?:, rpc latency p50 ([0-9.]+) us p99 ([0-9.]+) us)?", p.stdout)
        cons = re.search(r"conservation: (\d+) keys (\d+) decisions (\d+) violations", p.stdout)
        if not mm:
            out[label] = {"error": (p.stdout + p.stderr)[-400:]}
            continue
        out[label] = {"value": float(mm.group(1)) * 1e6, "unit": "decisions/s", "batches_per_s": float(mm.group(2)), "avg_batch": float(mm.group(3)),
                      "errors": int(mm.group(4)), "caller_threads": T, "shards": S, "items_per_rpc": items, "keys": keys, "api": api}
        wm = re.search(r"wire pool: (\d+) stages .*?per stage: first payload -> sealed ([0-9.]+) us, sealed -> decoded ([0-9.]+) us, decoded -> answers in host memory ([0-9.]+) us, "
                       r"([0-9.]+) items, ([0-9.]+) RPCs; the pool threads' own time per stage: decode enqueue ([0-9.]+) us, routing enqueue ([0-9.]+) us, evaluation enqueue ([0-9.]+) us", p.stdout)
        if wm:
            out[label]["per_stage"] = {"items": float(wm.group(5)), "rpcs": float(wm.group(6)), "fill_us": float(wm.group(2)), "decode_us": float(wm.group(3)),
                                       "evaluate_us": float(wm.group(4)), "host_enqueue_us": {"decode": float(wm.group(7)), "routing": float(wm.group(8)), "evaluation": float(wm.group(9))}}
        if mm.group(5):
            out[label]["rpc_latency_us"] = {"p50": float(mm.group(5)), "p99": float(mm.group(6))}
        # the gate of a number whose callers run concurrently (no serial order to replay): per-key conservation over every answer the pool gave
        if cons and int(cons.group(3)) == 0 and int(mm.group(4)) == 0:
            out[label]["parity"] = (f"per-key conservation over all {int(cons.group(2))} decisions on {int(cons.group(1))} keys since the pool's creation: admitted <= limit, the admitted "
                                    "hits' `remaining` are exactly {limit-1 .. limit-admitted} (count and sum), refused => remaining 0 and the window's tokens gone, limit echoed, "
                                    "no item error: 0 violations (tools/bench_pool.cpp)")
        else:
            out[label]["parity"] = "FAILED" if cons else "not checked"
            out[label].pop("value", None)
    head = out.get("rpc_1000", {})
    res = {"value": head.get("value"), "unit": "decisions/s"}
    res.update(out)
    gated = {k: v["value"] for k, v in out.items() if isinstance(v, dict) and v.get("value") and str(v.get("parity", "")).startswith("per-key conservation")}
    if gated:
        best = max(gated, key=gated.get)
        res["best"] = {"case": best, "value": gated[best], "unit": "decisions/s", "note": "the largest gated rate among the cases above (`value` stays rpc_1000 through guber_pool_get_rate_limits, as in earlier rounds)"}
    res["host_cpus_usable"] = quota if quota else os.cpu_count()
    res["wire"] = ("wire_*: caller threads x SERIALIZED GetRateLimitsReq messages through guber_wire_pool_get_rate_limits (include/guber_wire.h): per RPC the host does one "
                   "compare-and-swap and two memcpys (payload in, response out); decode (k_wire_*), HashKey, XXH64, placement, evaluation, the answers' order (guber_front) "
                   "and the marshalling of GetRateLimitsResp (k_wire_enc) on the device; the callers parse the response bytes inside the clock (that is what the "
                   "conservation gate reads)")
    res["workload"] = ("caller threads x RPCs through guber_pool_get_rate_limits (the C ABI a binding calls: structure-of-arrays in and out) -> GPUWorkerPool: "
                       "one dispatcher per device, fused launches over the shards' stages, placement on key hashes with online hot-key isolation, Zipf-1.1, "
                       "closed loop: front-end checks, HashKey, XXH64, placement, slot reservation, in-place stage filling, completion and response fan-out "
                       "included.  The callers' work is CPU-bound: host_cpus_usable is what the cgroup grants (cpu.max), not the core count")
    return res


def run_end_to_end(args, ctx, NOW0, seed):
    """host memory in, host memory out: guber_stage_* — requests written into device-visible host arrays, DMA brings them to
    HBM beside the previous batches' kernels, responses are written in place over PCIe, several batches in flight.  Every
    batch is a distinct part of the stream and every stage is used once per pass: `prefilled` = NP stages filled before the
    clock (what caller threads filling their own slots look like to the engine), `with_host_fill` = 4 stages refilled by this
    one thread (numpy) inside the clock."""
    import gubernator_amd as ga
    K, B = ctx.K, ctx.B
    rig = Rig(ctx, args.algo, args.dist, 1, duration_ms=args.duration_ms)
    rig.populate(NOW0)
    NP, NB = int(os.environ.get("GUBER_BENCH_E2E_PREFILLED", "160")), 288
    rig.warmup, rig.steps = 0, NP + NB
    rig.build_stream(NP + NB, NOW0, seed)
    eng = rig.engines[0]
    depth = max(1, int(os.environ.get("GUBER_BENCH_E2E_DEPTH", "3")))      # batches in flight
    bytes_per_req = (15 + 4 + 3 * 8 + 1 + 4) + 26                          # every request column present crosses PCIe once, responses once

    def pump(stages, nb, fill_from):
        t_sub, lat = {}, []
        h_submit = h_wait = 0.0
        ns = len(stages)
        t0 = time.perf_counter()
        for i in range(nb):
            st = stages[i % ns]
            if fill_from is not None:
                st.fill(fill_from[i])
            t_sub[i] = time.perf_counter()
            st.submit()
            h_submit += time.perf_counter() - t_sub[i]
            if i >= depth - 1:
                j = i - (depth - 1)
                tw = time.perf_counter()
                stages[j % ns].wait()
                h_wait += time.perf_counter() - tw
                lat.append((time.perf_counter() - t_sub[j]) * 1e6)
        for j in range(max(0, nb - depth + 1), nb):
            stages[j % ns].wait()
            lat.append((time.perf_counter() - t_sub[j]) * 1e6)
        el = time.perf_counter() - t0
        lat = sorted(lat[min(16, len(lat) // 4):])
        return {"value": round(nb * B / el, 1), "ms_per_step": round(el / nb * 1e3, 4), "batches": nb,
                "latency_us": {"p50": round(percentile(lat, 0.5), 1), "p99": round(percentile(lat, 0.99), 1), "n": len(lat)},
                "pcie_GBps": round(bytes_per_req * B * nb / el / 1e9, 2),
                "host_us_per_batch": {"submit": round(h_submit / nb * 1e6, 1), "wait": round(h_wait / nb * 1e6, 1)}}

    def make(n):
        ss = [ga.Stage(eng, B, key_bytes_cap=B * 16) for _ in range(n)]
        for st in ss:
            st.disable("burst", "created_at", "is_owner")
        return ss
    few = make(depth + 1)
    with_fill = pump(few, NB, [rig.host_batch(i) for i in range(NB)])          # also warms the path up
    got = {}                                                         # batch -> digest of the answers as they lie in the stage's HOST arrays
    for j in range(NB - (depth + 1), NB):                            # (the stages of this run are refilled under the clock: the last answers are still there)
        got[j] = host_digest(few[j % (depth + 1)].result())
    for st in few:
        st.close()
    many = make(NP)
    for k, st in enumerate(many):
        st.fill(rig.host_batch(NB + k))
    best = pump(many, NP, None)
    for k, st in enumerate(many):                                    # every batch of the headline run keeps its stage: all of them are checked
        got[NB + k] = host_digest(st.result())
    for st in many:
        st.close()
    parity = None
    if not args.no_cpu_baseline:
        import support
        w = min(os.cpu_count() or 1, 32)
        th = min(w, usable_cpus()) if w > 1 else 0
        orc = support.Oracle(cache_size=4 * K, workers=w)
        oracle_populate(rig, orc, th, NOW0)
        bad = [s for s in range(NB + NP) if (lambda want: s in got and host_digest(want) != got[s])(orc.eval(rig.host_batch(s), threads=th))]
        orc.close()
        parity = (f"bit-exact vs the oracle fed the whole sequence (populate, the {NB} batches of `with_host_fill`, the {NP} of the headline run, in order): {len(got)} batches by "
                  f"digest of the stages' HOST result arrays — all {NP} of the headline run, the last {depth + 1} of `with_host_fill` (its stages are refilled under the clock)"
                  if not bad else "FAILED")
        if bad:
            print(f"PARITY FAILURE: end_to_end batches {bad[:8]}", file=sys.stderr)
            best["value"] = None
    rig.close()
    return {"value": best["value"], "unit": "decisions/s", "ms_per_step": best["ms_per_step"], "steps": NP, "replays": 0, "parity": parity,
            "latency_us": best["latency_us"], "pcie_GBps": best["pcie_GBps"], "host_us_per_batch": best["host_us_per_batch"],
            "with_host_fill": with_fill,
            "workload": f"{K} keys, {args.dist}, {args.algo.upper()}_BUCKET, batch {B}, one table, guber_stage_submit / guber_stage_wait from one host thread: request "
                        "columns -> the stage's HBM mirror by DMA on a copy stream beside the previous batches' kernels, responses written straight into the host "
                        f"arrays by k_eval2, {depth} batches in flight; the headline line: {NP} distinct batches in {NP} stages filled before the clock, each used "
                        f"once; `with_host_fill`: {NB} further distinct batches through {depth + 1} stages, this thread's copy of every batch into the stage (numpy) included"}


def run_global(args, ctx, dist):
    """BASELINE config 5 on the NATIVE exchange: every request carries GLOBAL, every rank serves all keys from its replica,
    every K steps one guber_global_sync tick (guber_global_sync.h: hits -> owners over RCCL / device copies, owners apply and
    broadcast).  world > 1: one rank per process, guber_comm_create_rank (the unique id travels through torch.distributed);
    world == 1: --logical-ranks R engines on this GPU behind guber_comm_create_local (device-copy transport: RCCL refuses two
    ranks on one device)."""
    import torch
    import gubernator_amd as ga
    import streams
    from gubernator_amd import global_native as gn
    from gubernator_amd import shard
    K, B, world, rank, dev = ctx.K, ctx.B, ctx.world, ctx.rank, ctx.dev
    GSYNC, NOW0 = args.global_sync, streams.NOW0
    if world > 1 and args.one_device and not os.environ.get("GUBER_RCCL_LIB"):
        raise SystemExit("--global-sync with --one-device: RCCL refuses two ranks on one GPU; run ONE process with --logical-ranks R instead "
                         "(or rehearse the RCCL call sequence through the test-only librccl: GUBER_RCCL_LIB=tests/hostsim/libfake_rccl.so)")
    R = 1 if world > 1 else max(1, args.logical_ranks)          # ranks living in this process
    nranks = world if world > 1 else R
    ring = ga.Ring(shard.peer_names(nranks), 512, "fnv1")
    ctx.dispatch, ctx.streams = "threads", 1
    rigs = [Rig(ctx, args.algo, args.dist, 1, flags=ga.FLAG_GLOBAL, max_key_bytes=64, duration_ms=args.duration_ms) for _ in range(R)]
    engines = [r.engines[0] for r in rigs]
    resident = sum(r.populate(NOW0) for r in rigs)
    total_steps = args.warmup + args.steps
    owners = []
    for q, rig in enumerate(rigs):
        my_rank = rank if world > 1 else q
        rig.warmup, rig.steps, rig.profile_steps, rig.latency_steps = args.warmup, args.steps, 0, 0
        rig.build_stream(total_steps, NOW0, 1234 + my_rank * 64)
        rig.kept = {}
        d_owner = torch.empty(B, dtype=torch.int32, device=dev)   # ONE buffer, reused only after torch's stream has read it: a buffer per
        for s, b in enumerate(rig.batches):                     # batch went back to torch's allocator while the comparison below was still queued on
            # torch's stream, and the ENGINE's stream (which the allocator knows nothing about) wrote the next batch's owners into it — with two
            # processes sharing a GPU that race was lost in 4 runs of 10: a few hundred wrong is_owner flags, replicas that never converge
            rig.engines[0].route_dev(ring, b.key_bytes, b.key_off, B, d_owner.data_ptr())   # is_owner[i] = (ring owner of key i == this rank), on device
            t = (d_owner == my_rank).to(torch.uint8)
            torch.cuda.current_stream(dev).synchronize()
            owners.append(t)
            b.is_owner = t.data_ptr()
    torch.cuda.synchronize(dev)                                 # the owner flags are produced on torch's stream
    if world > 1:
        uid = [gn.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        comm = gn.Comm.rank(engines[0], rank, world, uid[0], ring)
    else:
        comm = gn.Comm.local(engines, ring, use_rccl=False)
    sync_stats = []

    def run(s):
        for rig in rigs:
            rig.engines[0].eval_dev(rig.batches[s], rig.scratch[0].c)
        if (s + 1) % GSYNC == 0:
            t_s = time.perf_counter()
            per = comm.sync(NOW0 + 1 + s)
            st = dict(comm.last)
            st["wall_ms"] = (time.perf_counter() - t_s) * 1e3
            st["per_rank"] = per
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
    # convergence as functional_test.go:1815-1821: after a final tick with nothing new in between, every replica reports the
    # same remaining for the keys of the last batch (hits = 0 reads)
    comm.sync(NOW0 + 1 + total_steps)
    comm.sync(NOW0 + 2 + total_steps)
    # (every replica is asked for the SAME keys: rank 0's last batch — each rank of an N > 1 run draws a stream of its own)
    L = rigs[0].L
    probe_keys = rigs[0].d_keys[(total_steps - 1) * B * L:(total_steps - 1) * B * L + B * L + 8].clone()
    if world > 1:
        pk = probe_keys.cpu() if args.backend != "nccl" else probe_keys
        dist.broadcast(pk, src=0)
        probe_keys = pk.to(dev)
        torch.cuda.synchronize(dev)
    reads = []
    for rig in rigs:
        b = rig.batch_struct(probe_keys.data_ptr(), B, NOW0 + 3 + total_steps, hits=0)
        res = rig.DevResult(rig, B)
        torch.cuda.synchronize(dev)
        rig.engines[0].eval_dev(b, res.c)
        rig.engines[0].synchronize()
        reads.append(res.remaining.clone())
    converged = all(bool(torch.equal(reads[0], r)) for r in reads[1:])
    if os.environ.get("GUBER_BENCH_GLOBAL_SUMS"):                # (diagnostics: the probe's answers as sums, per local replica)
        print(f"[global leg] rank {rank}: stream checksums {[int(r_.h_ids.astype(np.int64).sum()) for r_ in rigs]}, owner flags set {int(sum(int(t.sum()) for t in owners))}, "
              f"per sync (hits rows sent, applied, update rows, items installed): {[(x['hits_rows_sent'], x['hits_rows_applied'], x['update_rows'], x['items_installed']) for x in sync_stats]}",
              file=sys.stderr, flush=True)
        print(f"[global leg] rank {rank}: sums of the probe's remaining per local replica: {[int(r.sum()) for r in reads]}", file=sys.stderr, flush=True)
    if world > 1:
        on_host = args.backend != "nccl"                         # (gloo: host tensors)
        mine = reads[0].cpu() if on_host else reads[0]
        ref = mine.clone()
        dist.broadcast(ref, src=0)
        if not torch.equal(ref, mine):                            # what differs, for whoever reads the failure
            bad = (ref != mine).nonzero().flatten()
            print(f"[global leg] rank {rank}: {bad.numel()} of {B} reads differ from rank 0's; first: "
                  f"{[(int(i), int(mine[i]), int(ref[i])) for i in bad[:8]]}", file=sys.stderr, flush=True)
        flag = torch.tensor([1 if (converged and torch.equal(ref, mine)) else 0], device="cpu" if on_host else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        converged = bool(flag.item())
    fallbacks = shard.sum_over_ranks(int(sum(x["fallbacks"] for x in sync_stats)), device=dev if args.backend == "nccl" else None)
    if not converged:
        raise SystemExit("GLOBAL leg: the replicas did not converge after the final ticks (functional_test.go:1815-1821): refusing to report a number")
    if rank == 0:
        timed = sync_stats[args.warmup // GSYNC:] or sync_stats
        avg = lambda k: sum(x[k] for x in timed) / max(len(timed), 1)   # noqa: E731
        out = {"metric": "rate-limit decisions/sec (kernel path, inputs resident in HBM)", "value": round(args.steps * B * nranks / wall, 1),
               "unit": "decisions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(wall / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "int64" if args.algo == "token" else "f64", "data": "synthetic",
               "config": {"workload": f"{K} keys replicated on every rank, {args.dist} stream (distinct batches, never replayed), batch={B}, {args.algo.upper()}_BUCKET, GLOBAL behaviour, "
                                      f"guber_global_sync every {GSYNC} batches, {world}xMI355X" + (f", {R} logical ranks on one GPU" if world == 1 else ""),
                          "keys_per_gpu": K, "batch": B, "resident_items_local": int(resident), "ranks": nranks},
               "parity": (f"{nranks}/{nranks} ranks: after two final ticks every replica answers a hits=0 read of the last batch's {B} requests with the same remaining "
                          f"(functional_test.go:1815-1821); host fallbacks over all ranks: {fallbacks}"),
               "global_sync": {"every_batches": GSYNC, "syncs": len(sync_stats), "implementation": "native: guber_comm_* + guber_global_sync (guber_global_sync.h)",
                               "transport": (("the test-only librccl (GUBER_RCCL_LIB): ranks sharing one GPU" if os.environ.get("GUBER_RCCL_LIB") else "RCCL grouped send/recv (xGMI)") if world > 1 else "device copies between logical ranks of one GPU"),
                               "avg_ms": round(avg("ms"), 3), "avg_wall_ms": round(avg("wall_ms"), 3),
                               "avg_hits_rows_sent": int(avg("hits_rows_sent")), "avg_hits_rows_applied": int(avg("hits_rows_applied")),
                               "avg_update_rows": int(avg("update_rows")), "avg_items_installed": int(avg("items_installed")),
                               "avg_bytes_moved": int(avg("bytes_moved")), "host_fallbacks": int(sum(x["fallbacks"] for x in sync_stats)),
                               "replicas_converged": converged,
                               # the answers themselves, as one number: the streams are seeded, so every run of the same flags — as two
                               # processes or as two logical ranks of one — must say the same (tests/test_gpu_bench_multi.py)
                               "probe_remaining_sum": int(reads[0].sum())}}
        print(json.dumps(out), flush=True)
    comm.close()
    for rig in rigs:
        rig.close()
    if world > 1:
        finish_distributed(dist)


if __name__ == "__main__":
    main()
