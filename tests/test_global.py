"""GLOBAL behaviour (BASELINE config 5) on the CPU:
 * the reference model (tests/global_model.py, global.go restated over N oracles) reproduces the GLOBAL
   vectors of the reference's functional tests;
 * the product orchestrator (tests/pyglobal.py GlobalSync) driven with oracle-backed nodes gives
   the same answers as the model, in-process (LocalCluster) and across 2 gloo ranks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import gubernator_amd as ga
import scenarios
import support
from global_model import GlobalModel, OracleNode
import pyglobal as global_sync
from support import HostBatch

HERE = os.path.dirname(os.path.abspath(__file__))
NOW = 1_700_000_000_000


def pick_key(ring, owner_rank, prefix):
    for i in range(10_000):
        k = f"{prefix}_{i}"
        if int(ring.route([k])[0]) == owner_rank:
            return k
    raise AssertionError


def run_vectors(request_fn, sync_fn, ring, n_peers):
    n = 0
    for sc in scenarios.load("global_vectors.json")["scenarios"]:
        key = pick_key(ring, 0, sc["name"]).encode()       # rank 0 owns the key; p<i> = rank i+1
        reset0 = None
        now = NOW
        for st in sc["steps"]:
            rank = 0 if st["peer"] == "o" else 1 + int(st["peer"][1:])
            assert rank < n_peers
            req = dict(key=key, hits=st["hits"], limit=sc["limit"], duration=sc["duration"], algorithm=sc["algorithm"],
                       behavior=st["behavior"], burst=0)
            status, limit, remaining, reset_time, err = request_fn(rank, req, now)
            where = f"{sc['name']} {st}"
            assert err == 0 and limit == sc["limit"], where
            if "status" in st["expect"]:
                assert status == st["expect"]["status"], where
            if "remaining" in st["expect"]:
                assert remaining == st["expect"]["remaining"], where
            if sc.get("reset_time_constant"):
                reset0 = reset0 or reset_time
                assert reset_time == reset0, where
            n += 1
            now += 3
            if st["sync_after"]:
                stats = sync_fn(now)
                es = st.get("expect_sync")
                if es and isinstance(stats, list) and stats and isinstance(stats[0], dict):   # per-rank stats of the orchestrator
                    rk = lambda pr: 0 if pr == "o" else 1 + int(pr[1:])
                    senders = sorted(r for r, s_ in enumerate(stats) if s_["hits_sent"] > 0)
                    assert senders == sorted(rk(pr) for pr in es["hits_from"]), (where, stats)
                    assert all(stats[r]["hits_sent"] == 1 for r in senders), (where, stats)        # one aggregated row per sender and key
                    casters = sorted(r for r, s_ in enumerate(stats) if s_["broadcast"] > 0)
                    assert casters == sorted(rk(pr) for pr in es["broadcast_from"]) and stats[0]["broadcast"] == 1, (where, stats)
                    n += 1
    return n


def test_model_reproduces_reference_global_vectors():
    ring = ga.Ring([f"gpu{i}" for i in range(6)])
    def fresh():
        m = GlobalModel(6, lambda k: int(ring.route([k])[0]))
        return m
    m = fresh()
    assert run_vectors(m.request, m.sync, ring, 6) >= 40


def cluster_request(cluster, rank, req, now):
    gs = cluster.ranks[rank]
    res = gs.evaluate([req["key"]], req["hits"], req["limit"], req["duration"], now, burst=req["burst"],
                      algorithm=req["algorithm"], behavior=req["behavior"], created_at=now)
    return res.rows()[0]


def test_orchestrator_with_oracle_nodes_reproduces_vectors():
    ring = ga.Ring([f"gpu{i}" for i in range(6)])
    cluster = global_sync.LocalCluster([OracleNode() for _ in range(6)], ring)
    assert run_vectors(lambda r, q, now: cluster_request(cluster, r, q, now), cluster.sync, ring, 6) >= 40


def random_global_stream(seed, n_ranks, steps, keys=40):
    rng = np.random.default_rng(seed)
    now = NOW
    for s in range(steps):
        rank = int(rng.integers(0, n_ranks))
        n = int(rng.integers(1, 60))
        ids = rng.integers(0, keys, n)
        batch = dict(keys=[f"glob_{int(i)}".encode() for i in ids],
                     hits=rng.choice([0, 1, 1, 2, 5, -1], n), limit=rng.choice([5, 20, 20, 100], n),
                     duration=rng.choice([50, 1000, 60000], n), algorithm=(ids % 2).astype(np.uint8),
                     behavior=np.where(rng.random(n) < 0.05, 8, 0).astype(np.uint32))
        yield rank, batch, now, rng.random() < 0.3
        now += int(rng.choice([0, 1, 5, 40]))


def run_random(cluster_eval, cluster_sync, model, n_ranks, seed, steps=150):
    for rank, b, now, do_sync in random_global_stream(seed, n_ranks, steps):
        got = cluster_eval(rank, b, now)
        want = [model.request(rank, dict(key=k, hits=int(h), limit=int(l), duration=int(d), algorithm=int(a),
                                         behavior=int(bh) | 2, burst=0), now)
                for k, h, l, d, a, bh in zip(b["keys"], b["hits"], b["limit"], b["duration"], b["algorithm"], b["behavior"])]
        assert got.rows() == want, (seed, rank, now)
        if do_sync:
            cluster_sync(now)
            model.sync(now)


@pytest.mark.parametrize("seed", [1, 2])
def test_orchestrator_matches_model_on_random_streams(seed):
    n = 4
    ring = ga.Ring([f"gpu{i}" for i in range(n)])
    cluster = global_sync.LocalCluster([OracleNode() for _ in range(n)], ring)
    model = GlobalModel(n, lambda k: int(ring.route([k])[0]))
    run_random(lambda r, b, now: cluster.ranks[r].evaluate(b["keys"], b["hits"], b["limit"], b["duration"], now,
                                                           algorithm=b["algorithm"], behavior=b["behavior"], burst=0,
                                                           created_at=now),
               cluster.sync, model, n, seed)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _gloo_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ring = ga.Ring([f"gpu{i}" for i in range(world)])
    gs = global_sync.GlobalSync(OracleNode(), rank, world, ring, global_sync.TorchTransport())
    model = GlobalModel(world, lambda k: int(ring.route([k])[0]))   # every rank replays the whole model
    for r, b, now, do_sync in random_global_stream(7, world, 120):
        want = [model.request(r, dict(key=k, hits=int(h), limit=int(l), duration=int(d), algorithm=int(a),
                                      behavior=int(bh) | 2, burst=0), now)
                for k, h, l, d, a, bh in zip(b["keys"], b["hits"], b["limit"], b["duration"], b["algorithm"], b["behavior"])]
        if r == rank:
            got = gs.evaluate(b["keys"], b["hits"], b["limit"], b["duration"], now, algorithm=b["algorithm"],
                              behavior=b["behavior"], burst=0, created_at=now)
            assert got.rows() == want, (rank, now)
        if do_sync:
            gs.sync(now)
            model.sync(now)
    dist.barrier()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


def test_global_sync_over_gloo_two_ranks(tmp_path):
    port = _free_port()
    mp.spawn(_gloo_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(2))


def _gloo_transport_worker(rank, world, port, out_dir):
    """The row collectives of the device-resident exchange (pyglobal_dev.TorchTransportDev): all_to_all_single with
    variable splits and a padded all_gather_into_tensor, here over gloo on CPU tensors (RCCL on the GPU)."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    import pyglobal_dev as gsd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = gsd.TorchTransportDev("cpu")
    rb = 12
    for it in range(6):
        rng = np.random.default_rng(100 + it)                         # same stream on every rank
        counts = rng.integers(0, 5, (world, world))                   # counts[src][dst]
        rows = {src: rng.integers(0, 256, (int(counts[src].sum()), rb), dtype=np.uint8) for src in range(world)}
        recv = t.exchange_rows(torch.from_numpy(rows[rank].copy()), [int(c) for c in counts[rank]])
        want = np.concatenate([rows[src][int(counts[src][:rank].sum()):int(counts[src][:rank + 1].sum())] for src in range(world)])
        assert np.array_equal(recv.numpy(), want), (rank, it)
        ns = rng.integers(0, 4, world)
        items = {src: rng.integers(0, 256, (int(ns[src]), rb), dtype=np.uint8) for src in range(world)}
        got = t.gather_rows(torch.from_numpy(items[rank].copy()))
        assert len(got) == world and all(np.array_equal(got[src].numpy(), items[src]) for src in range(world)), (rank, it)
    dist.barrier()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_device_exchange_collectives_over_gloo(tmp_path, world):
    port = _free_port()
    mp.spawn(_gloo_transport_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))
