#!/usr/bin/env python3
"""bench.py — rate-limit decisions/sec of the MI355X engine on BASELINE.json's workload.

One "step" = one GetRateLimits batch of 65536 checks evaluated by the HIP path
(guber_eval_batch_dev: k_resolve -> radix passes -> k_heads -> k_eval) with every input array
already resident in HBM.  Workload (BASELINE.json configs[1]): 10M resident
keys per GPU, Zipf(1.1) key popularity (stream seed 1234, permutation seed 99), TOKEN_BUCKET, hits 1,
limit 100, duration 60 s, now_ms advancing 1 ms per batch.  `--algo leaky` switches to configs[2].

Inside a GPU the resident keys are split into S logical shards (default 4; the reference shards its key
space the same way over Config.Workers goroutines, workers.go:19-25): S engines with their own HBM tables
and HIP streams, the front end routes a key to its shard with the same consistent hash, and step s
evaluates one 65536-request batch of shard s % S — consecutive steps are independent and overlap on the
GPU.  `--shards 1` gives the single-table number (2.1 G/s vs 3.5 G/s on MI355X).

N > 1 (launched by torch.distributed.run, one rank per GPU): the key space is N x 10M keys sharded by
the reference's replicated consistent hash (replicated_hash.go; 512 vnodes, fnv1, peers gpu0..gpuN-1),
every rank evaluates batches over the keys it owns — no data-path collective (weak scaling).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
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
KERNEL_BYTES = {"token": {"k_front": 76, "k_eval2": 73, "k_resolve": 28, "k_eval": 121},
                "leaky": {"k_front": 84, "k_eval2": 89, "k_resolve": 28, "k_eval": 145}}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--keys", type=int, default=10_000_000, help="resident keys per GPU")
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--algo", choices=["token", "leaky"], default="token")
    ap.add_argument("--dist", choices=["zipf", "uniform"], default="zipf")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batches", type=int, default=48)
    ap.add_argument("--cpu-threads", type=int, default=32, help="worker shards/threads of the CPU baseline")
    ap.add_argument("--profile-steps", type=int, default=32)
    ap.add_argument("--shards", type=int, default=4, metavar="S",
                    help="logical key-space shards per GPU (the reference's Config.Workers sharding, workers.go:19-25): S "
                         "engines with their own tables and streams; step s evaluates a batch of shard s %% S")
    ap.add_argument("--global-host", action="store_true", help="with --global-sync: use the host-staged exchange (global_sync.py)")
    ap.add_argument("--global-sync", type=int, default=0, metavar="K",
                    help="BASELINE config 5: every request carries GLOBAL, every rank serves ALL keys from its replica, "
                         "and every K steps the ranks exchange pending hits / broadcast owner state (0 = off)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL)")
    ap.add_argument("--one-device", action="store_true",
                    help="debug: all ranks share GPU 0 (single-GPU box; use with --backend gloo)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import gubernator_amd as ga
    import streams

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
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

    K, B = args.keys, args.batch
    algo_id = 0 if args.algo == "token" else 1
    stream = torch.cuda.Stream(device=dev)
    GSYNC = args.global_sync
    S = max(1, args.shards)
    if GSYNC:
        S = 1   # GLOBAL replicas: one table per GPU
    sstreams = [stream] + [torch.cuda.Stream(device=dev) for _ in range(S - 1)]
    engines = [ga.Engine(cache_size=(K + K // 4) // S + 1024, device=local_rank, max_batch=B, stream=sstreams[j].cuda_stream,
                         max_key_bytes=64 if GSYNC else 0, flags=ga.FLAG_GLOBAL if GSYNC else 0) for j in range(S)]
    eng = engines[0]

    # ---- key ownership: ids of the global key space (world x K) this rank owns on the ring ------
    total_keys = K if GSYNC else K * world      # GLOBAL: one key space, replicated on every GPU
    table = streams.key_table(total_keys)
    from gubernator_amd import shard

    def route_on_device(ring, kb, ko):       # ReplicatedConsistentHash.Get for a chunk of keys (k_route)
        d_kb, d_ko = torch.from_numpy(kb).to(dev), torch.from_numpy(ko.view(np.int32)).to(dev)
        d_owner = torch.empty(len(ko) - 1, dtype=torch.int32, device=dev)
        eng.route_dev(ring, d_kb.data_ptr(), d_ko.data_ptr(), len(ko) - 1, d_owner.data_ptr())
        return d_owner.cpu().numpy()

    if GSYNC:
        my_ids = np.arange(total_keys)          # every rank holds (a replica of) every key
        ring = ga.Ring(shard.peer_names(world), 512, "fnv1")
    else:
        my_ids = shard.owned_key_ids(table, world, rank, route=route_on_device, chunk=4_000_000)
    nk = len(my_ids)
    # logical shards inside this GPU: the rank's keys are split once more by the same kind of ring
    if S > 1:
        sring = ga.Ring([f"gpu{rank}-shard{j}" for j in range(S)], 512, "fnv1")
        sown = np.concatenate([route_on_device(sring, *streams.keys_for_ids(table, my_ids[lo:lo + 4_000_000]))
                               for lo in range(0, nk, 4_000_000)])
        shard_ids = [my_ids[sown == j] for j in range(S)]
    else:
        shard_ids = [my_ids]

    # ---- device-resident batches ------------------------------------------------------------
    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    class DevBatch:
        def __init__(self, ids, now_ms, hits=1):
            kb, ko = streams.keys_for_ids(table, ids)
            n = len(ids)
            self.n = n
            self.t = [to_dev(kb), to_dev(ko.view(np.int32)),
                      torch.full((n,), hits, dtype=torch.int64, device=dev),
                      torch.full((n,), 100, dtype=torch.int64, device=dev),
                      torch.full((n,), 60_000, dtype=torch.int64, device=dev),
                      torch.full((n,), algo_id, dtype=torch.uint8, device=dev),
                      torch.full((n,), 2 if GSYNC else 0, dtype=torch.int32, device=dev)]
            p = [x.data_ptr() for x in self.t]
            owner_ptr = None
            if GSYNC and world > 1:                 # is_owner[i] = (ring owner of key i == this rank), on device
                d_owner = torch.empty(n, dtype=torch.int32, device=dev)
                eng.route_dev(ring, p[0], p[1], n, d_owner.data_ptr())
                self.t.append((d_owner == rank).to(torch.uint8))
                owner_ptr = self.t[-1].data_ptr()
            self.c = ga.GuberBatch(n, 0, p[0], p[1], p[2], p[3], p[4], None, None, p[5], p[6], owner_ptr, None, None,
                                   int(now_ms))

    class DevResult:
        def __init__(self, n):
            self.status = torch.empty(n, dtype=torch.uint8, device=dev)
            self.err = torch.empty(n, dtype=torch.uint8, device=dev)
            self.limit = torch.empty(n, dtype=torch.int64, device=dev)
            self.remaining = torch.empty(n, dtype=torch.int64, device=dev)
            self.reset_time = torch.empty(n, dtype=torch.int64, device=dev)
            self.c = ga.GuberResult(self.status.data_ptr(), self.limit.data_ptr(), self.remaining.data_ptr(),
                                    self.reset_time.data_ptr(), self.err.data_ptr(), 0, 0, 0, 0, 0)

        def host(self):
            h = ga.HostResult(len(self.status))
            for name in ("status", "limit", "remaining", "reset_time", "err"):
                getattr(h, name)[:] = getattr(self, name).cpu().numpy()
            return h

    NOW0 = streams.NOW0
    # residency: every owned key gets a bucket before anything is timed (hits 0 = create, consume nothing)
    scratches = [DevResult(B) for _ in range(S)]
    scratch = scratches[0]
    for j in range(S):
        with torch.cuda.stream(sstreams[j]):
            for lo in range(0, len(shard_ids[j]), B):
                db = DevBatch(shard_ids[j][lo:lo + B], NOW0, hits=0)
                engines[j].eval_dev(db.c, scratches[j].c)
                engines[j].synchronize()
    resident = sum(e_.size() for e_ in engines)

    draws = []
    for j in range(S):
        ids_j = shard_ids[j]
        if args.dist == "zipf":
            smp = streams.ZipfSampler(len(ids_j), s=1.1, seed=1234 + rank * 64 + j, perm_seed=99)
            draws.append(lambda n, smp=smp, ids_j=ids_j: ids_j[smp.draw(n)])
        else:
            rg = np.random.default_rng(1234 + rank * 64 + j)
            draws.append(lambda n, rg=rg, ids_j=ids_j: ids_j[rg.permutation(len(ids_j))[:n]] if n <= len(ids_j) else ids_j[rg.integers(0, len(ids_j), n)])

    total_steps = args.warmup + args.steps
    host_ids = [draws[s % S](B) for s in range(total_steps)]
    batches = [DevBatch(host_ids[s], NOW0 + 1 + s) for s in range(total_steps)]
    KEEP = min(8, total_steps)           # results of the first KEEP steps are kept for the parity gate
    kept = [DevResult(B) for _ in range(KEEP)]

    gsync = None
    if GSYNC:
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
        engines[s % S].eval_dev(batches[s].c, (kept[s] if s < KEEP else scratches[s % S]).c)
        if gsync is not None and (s + 1) % GSYNC == 0:
            t_s = time.perf_counter()
            with torch.cuda.stream(stream):          # the engine's stream: torch ops of the exchange and engine kernels stay ordered
                st = gsync.sync(NOW0 + 1 + s)
                stream.synchronize()
            st["ms"] = (time.perf_counter() - t_s) * 1e3
            sync_stats.append(st)

    def barrier():
        if world > 1:
            dist.barrier()

    import threading

    def run_range(lo, hi, j):
        # steps of shard j in [lo, hi): one driver thread per shard, as one batcher goroutine per shard would
        for s_ in range(lo, hi):
            if s_ % S == j:
                run(s_)

    def run_steps(lo, hi):
        if S == 1 or gsync is not None:
            for s_ in range(lo, hi):
                run(s_)
            return
        ts = [threading.Thread(target=run_range, args=(lo, hi, j)) for j in range(S)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    run_steps(0, args.warmup)
    torch.cuda.synchronize(dev)
    barrier()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
    t0 = time.perf_counter()
    for j in range(S):
        ev0[j].record(sstreams[j])
    run_steps(args.warmup, total_steps)
    for j in range(S):
        ev1[j].record(sstreams[j])
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    barrier()
    wall = t1 - t0
    ev_ms = max(ev0[j].elapsed_time(ev1[j]) for j in range(S))
    wall = shard.max_over_ranks(wall, device=red_dev)
    decisions = args.steps * B * world
    value = decisions / wall

    # ---- per-kernel durations (HIP events on the engine stream), same inputs ---------------------
    roofline = None
    kernel_ms = {}
    if rank == 0 and args.profile_steps > 0:
        shard0_steps = [s_ for s_ in range(args.warmup, total_steps) if s_ % S == 0]
        eng.profile(True)
        eng.profile_read()
        with torch.cuda.stream(stream):
            for j in range(args.profile_steps):
                eng.eval_dev(batches[shard0_steps[j % len(shard0_steps)]].c, scratch.c)
        prof = eng.profile_read()
        eng.profile(False)
        kernel_ms = {k: (ms / n if n else 0.0) for k, (n, ms) in prof.items()}
        cand = {k: v for k, v in kernel_ms.items() if k in KERNEL_BYTES[args.algo] and v > 0}
        dom = max(cand, key=cand.get)
        dom_bytes = KERNEL_BYTES[args.algo][dom] * B
        achieved = dom_bytes / (cand[dom] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.algo, {}).get(dom)
            except Exception:
                traffic = None
        step_ms_events = ev_ms / args.steps
        # measured ceilings of this GPU (tools/hbm_peak.hip -> profiles/hbm_peak.json) and the HBM traffic of one batch
        # from the PMC passes (profiles/roofline_traffic.json): how close the pipeline runs to what random access allows
        measured = None
        try:
            hp = json.load(open(os.path.join(ROOT, "profiles", "hbm_peak.json")))
            tj = json.load(open(tpath))
            cor = sum(tj.get(args.algo, {}).get(k, 0) for k in KERNEL_BYTES[args.algo])
            raw = sum(tj.get(args.algo + "_raw", {}).get(k, 0) for k in KERNEL_BYTES[args.algo])
            if cor and raw:
                g_raw, g_cor = (x / (step_ms_events * 1e-3) / 1e9 for x in (raw, cor))
                measured = {"stream_read_GBps": hp["stream_read_GBps"], "random_gather_GBps": hp["random_gather_128B_GBps"],
                            "hbm_traffic_bytes_per_batch": {"raw": raw, "corrected": cor},
                            "hbm_traffic_GBps": {"raw": round(g_raw, 1), "corrected": round(g_cor, 1)},
                            "frac_of_random_gather": {"raw": round(g_raw / hp["random_gather_128B_GBps"], 4),
                                                      "corrected": round(g_cor / hp["random_gather_128B_GBps"], 4)}}
        except Exception:
            measured = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic,
                    "algorithmic_bytes_per_launch": dom_bytes,
                    "kernel_avg_us": {k: round(v * 1e3, 2) for k, v in kernel_ms.items()},
                    "measured_ceilings": measured,
                    "pipeline": {"bytes_per_decision": BYTES_PER_DECISION[args.algo],
                                 "ms_per_batch_events": round(step_ms_events, 5),
                                 "achieved": round(BYTES_PER_DECISION[args.algo] * B / (step_ms_events * 1e-3) / 1e9, 2),
                                 "frac": round(BYTES_PER_DECISION[args.algo] * B / (step_ms_events * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6)}}

    # ---- single-batch latency: submit -> complete, one batch in flight (BASELINE metric: p99 batch latency) ----
    latency = None
    if rank == 0 and not GSYNC:
        lat = []
        shard0 = [s_ for s_ in range(args.warmup, total_steps) if s_ % S == 0]
        with torch.cuda.stream(stream):
            for j in range(min(200, 4 * len(shard0))):
                a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                eng.eval_dev(batches[shard0[j % len(shard0)]].c, scratch.c)
                b_.record(stream)
                b_.synchronize()
                lat.append(a.elapsed_time(b_) * 1e3)
        lat.sort()
        latency = {"unit": "us", "p50": round(lat[len(lat) // 2], 2), "p99": round(lat[min(len(lat) - 1, int(len(lat) * 0.99))], 2),
                   "min": round(lat[0], 2), "n": len(lat), "what": "one 65536-request batch, HIP events around guber_eval_batch_dev, nothing else in flight"}

    # ---- CPU baseline + parity gate (rank 0, N = 1 only) -------------------------------------------
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not GSYNC:
        import support
        threads = max(1, min(os.cpu_count() or 1, args.cpu_threads))
        orc = support.Oracle(cache_size=4 * K, workers=threads)
        for lo in range(0, nk, 1 << 18):
            orc.eval(streams.bench_batch(table, my_ids[lo:lo + (1 << 18)], NOW0, hits=0, algorithm=algo_id), threads=threads)
        nb = min(args.cpu_batches, total_steps)
        hb = [streams.bench_batch(table, host_ids[s], NOW0 + 1 + s, algorithm=algo_id) for s in range(nb)]
        # parity gate: the GPU's answers for the first KEEP batches of this very stream
        ok = True
        c0 = time.perf_counter()
        outs = [orc.eval(hb[s], threads=threads) for s in range(nb)]
        c1 = time.perf_counter()
        for s in range(KEEP):
            try:
                support.assert_results_equal(kept[s].host(), outs[s], f"bench batch {s}")
            except AssertionError as ex:
                ok = False
                print("PARITY FAILURE:", ex, file=sys.stderr)
        parity = "bit-exact vs oracle on the first %d batches" % KEEP if ok else "FAILED"
        mt = nb * B / (c1 - c0)
        # single-thread leg on a fresh, smaller sample of the same stream
        nb1 = max(4, nb // 4)
        c0 = time.perf_counter()
        for s in range(nb1):
            orc.eval(hb[s])
        c1 = time.perf_counter()
        st = nb1 * B / (c1 - c0)
        cpu = {"value": round(mt, 1), "unit": "decisions/s", "cores": threads, "kind": "port",
               "sample": f"{nb} batches of {B} from the same stream, {K} resident keys, oracle in the reference's "
                         f"worker-sharded design ({threads} workers/threads); single thread: {round(st, 1)} decisions/s "
                         f"over {nb1} batches",
               "single_thread_value": round(st, 1)}
        if not ok:
            raise SystemExit("parity gate failed: refusing to report a number")

    if rank == 0:
        out = {
            "metric": "rate-limit decisions/sec (kernel path, inputs resident in HBM)",
            "value": round(value, 1), "unit": "decisions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "int64" if args.algo == "token" else "f64", "data": "synthetic",
            "config": {"workload": f"{K} resident keys per GPU, {args.dist} key popularity"
                                   + (" s=1.1" if args.dist == "zipf" else "") +
                                   f", batch={B}, {args.algo.upper()}_BUCKET, hits=1 limit=100 duration=60000ms, "
                                   f"{world}xMI355X" + (", keys sharded by replicated consistent hash (512 vnodes, fnv1)" if world > 1 else "")
                                   + (f", {S} logical shards per GPU (own table + stream each, batches routed by the same hash)" if S > 1 else ""),
                       "keys_per_gpu": K, "batch": B, "algorithm": args.algo, "resident_items_rank0": int(resident),
                       "logical_shards_per_gpu": S,
                       "host_cores": os.cpu_count()},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "batch_latency": latency,
        }
        if GSYNC:
            timed = sync_stats[args.warmup // GSYNC:] or sync_stats
            out["config"]["workload"] += f", GLOBAL behaviour, sync every {GSYNC} batches"
            out["global_sync"] = {"every_batches": GSYNC, "syncs": len(sync_stats),
                                  "avg_ms": round(sum(x["ms"] for x in timed) / max(len(timed), 1), 3),
                                  "avg_rows_broadcast": int(sum(x["broadcast"] for x in timed) / max(len(timed), 1)),
                                  "avg_hits_rows_sent": int(sum(x["hits_sent"] for x in timed) / max(len(timed), 1)),
                                  "bytes_moved_rank0": gsync.bytes_moved,
                                  "exchange": "host-staged" if args.global_host else "device-resident (RCCL on HBM rows)"}
        print(json.dumps(out))
    for e_ in engines:
        e_.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
