"""The RCCL branch of guber_global_sync — grouped ncclSend / ncclRecv pairs, the count exchange by ncclAllGather,
guber_comm_create_rank / ncclCommInitRank, ncclCommInitAll for the ranks of one process — EXECUTED, on a box with one GPU.

RCCL itself refuses two ranks on one device, so on the 1-GPU boxes this repository is tested on that branch never ran (the
device-copy transport did).  tests/hostsim/fake_rccl.cpp is a test-only librccl with RCCL's signatures and group semantics for
ranks that SHARE a GPU (device -> POSIX shared memory -> device); the product loads it through GUBER_RCCL_LIB like any librccl.
What runs through it is the product's own call sequence — the code an 8-GPU node will run over xGMI:
  * one process, six logical ranks (ncclCommInitAll): the reference's 13 GLOBAL scenarios (functional_test.go:959-1341,
    1690-2097) incl. who sends hits and who broadcasts, and random streams against the global.go model;
  * six PROCESSES, one rank each (ncclCommInitRank, the unique id travelling through gloo): the 13 scenarios;
  * two processes: a random stream against the model, replicas converging.
Each leg asserts that Send / Recv calls really went through the stand-in."""
import ctypes
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FAKE = os.path.join(HERE, "hostsim", "libfake_rccl.so")


def _build():
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "hostsim"), "fake_rccl"], check=True)


def _fake_calls():
    L = ctypes.CDLL(FAKE)
    L.fake_rccl_total_calls.restype = ctypes.c_ulonglong
    return L.fake_rccl_total_calls()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


# ---- one process, logical ranks (runs in a child interpreter: the product resolves its RCCL once per process) -----------------
def _local_leg():
    sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
    import gubernator_amd as ga
    import test_global as tg
    from global_model import GlobalModel
    from gubernator_amd import global_native as gn
    mk = lambda: ga.Engine(cache_size=4096, max_batch=4096, max_key_bytes=64, flags=ga.FLAG_GLOBAL)
    ring = ga.Ring([f"gpu{i}" for i in range(6)])
    cluster = gn.Comm.local([mk() for _ in range(6)], ring, use_rccl=True)
    assert tg.run_vectors(lambda r, q, now: tg.cluster_request(cluster, r, q, now), cluster.sync, ring, 6) >= 40
    cluster.close()
    assert _fake_calls() > 0, "the RCCL branch did not run"
    n = 4
    for seed in (1, 2):
        ring4 = ga.Ring([f"gpu{i}" for i in range(n)])
        cl = gn.Comm.local([mk() for _ in range(n)], ring4, use_rccl=True)
        model = GlobalModel(n, lambda k: int(ring4.route([k])[0]))
        tg.run_random(lambda r, b, now: cl.ranks[r].evaluate(b["keys"], b["hits"], b["limit"], b["duration"], now, algorithm=b["algorithm"],
                                                             behavior=b["behavior"], burst=0, created_at=now),
                      cl.sync, model, n, seed, steps=150)
        cl.sync(tg.NOW + 10_000); model.sync(tg.NOW + 10_000)
        for k in range(40):
            key = f"glob_{k}".encode()
            vals = {r: (cl.ranks[r].node.get_item(key, tg.NOW + 10_000) or {}).get("remaining") for r in range(n)}
            want = {r: (model.oracles[r].get_item(key, tg.NOW + 10_000) or {}).get("remaining") for r in range(n)}
            assert vals == want, (seed, key, vals, want)
        assert cl.last["fallbacks"] == 0
        cl.close()
    print("local leg ok, Send/Recv calls through the stand-in:", _fake_calls())


def test_logical_ranks_of_one_process_through_the_rccl_calls():
    _build()
    env = dict(os.environ, GUBER_RCCL_LIB=FAKE)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "local"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "local leg ok" in r.stdout


# ---- one rank per process ---------------------------------------------------------------------------------------------------
def _vectors_on_my_rank(rank, comm, ring, n_peers):
    """test_global.run_vectors for a world whose ranks are processes: everybody walks the same script, the rank a step names
    evaluates and checks it, the sync is the collective"""
    import scenarios
    import test_global as tg
    n = 0
    for sc in scenarios.load("global_vectors.json")["scenarios"]:
        key = tg.pick_key(ring, 0, sc["name"]).encode()
        reset0, now = None, tg.NOW
        for st in sc["steps"]:
            r = 0 if st["peer"] == "o" else 1 + int(st["peer"][1:])
            assert r < n_peers
            if r == rank:
                req = dict(key=key, hits=st["hits"], limit=sc["limit"], duration=sc["duration"], algorithm=sc["algorithm"], behavior=st["behavior"], burst=0)
                status, limit, remaining, reset_time, err = tg.cluster_request(comm, 0, req, now)
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
                mine = comm.sync(now)[0]
                es = st.get("expect_sync")
                if es:                                         # who sends hits, who broadcasts (global.go:144-233, 234-283)
                    rk = lambda pr: 0 if pr == "o" else 1 + int(pr[1:])
                    assert (mine["hits_sent"] > 0) == (rank in [rk(pr) for pr in es["hits_from"]]), (sc["name"], st, rank, mine)
                    assert (mine["broadcast"] > 0) == (rank in [rk(pr) for pr in es["broadcast_from"]]), (sc["name"], st, rank, mine)
    return n


def _worker(rank, world, port, out_dir, mode):
    sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), GUBER_RCCL_LIB=FAKE)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gubernator_amd as ga
    import test_global as tg
    from global_model import GlobalModel
    from gubernator_amd import global_native as gn
    ring = ga.Ring([f"gpu{i}" for i in range(world)])
    node = ga.Engine(cache_size=4096, device=0, max_batch=4096, max_key_bytes=64, flags=ga.FLAG_GLOBAL)
    uid = [gn.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    comm = gn.Comm.rank(node, rank, world, uid[0], ring)                       # ncclCommInitRank on every process
    if mode == "vectors":
        done = _vectors_on_my_rank(rank, comm, ring, world)
        note = f"{done} steps evaluated on this rank"
    else:
        model = GlobalModel(world, lambda k: int(ring.route([k])[0]))          # every rank replays the whole model
        for r, b, t, do_sync in tg.random_global_stream(11, world, 150):
            want = [model.request(r, dict(key=k, hits=int(h), limit=int(l), duration=int(d), algorithm=int(a), behavior=int(bh) | 2, burst=0), t)
                    for k, h, l, d, a, bh in zip(b["keys"], b["hits"], b["limit"], b["duration"], b["algorithm"], b["behavior"])]
            if r == rank:
                got = comm.ranks[0].evaluate(b["keys"], b["hits"], b["limit"], b["duration"], t, algorithm=b["algorithm"], behavior=b["behavior"],
                                             burst=0, created_at=t)
                assert got.rows() == want, (rank, t)
            if do_sync:
                comm.sync(t)
                model.sync(t)
        comm.sync(tg.NOW + 10_000); model.sync(tg.NOW + 10_000)
        for k in range(40):                                                    # the replicas converged on the model's state
            key = f"glob_{k}".encode()
            got = (node.get_item(key, tg.NOW + 10_000) or {}).get("remaining")
            want = (model.oracles[rank].get_item(key, tg.NOW + 10_000) or {}).get("remaining")
            assert got == want, (rank, key, got, want)
        assert comm.last["fallbacks"] == 0
        note = "random stream ok"
    calls = _fake_calls()
    assert calls > 0, "the RCCL branch did not run"
    comm.close()
    node.close()
    dist.barrier()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write(f"{note}; {calls} Send/Recv calls through the stand-in")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(6, "vectors"), (2, "random")])
def test_one_rank_per_process_through_the_rccl_calls(tmp_path, world, mode):
    import torch.multiprocessing as mp
    _build()
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), mode), nprocs=world, join=True)
    notes = [open(tmp_path / f"ok{r}").read() for r in range(world)]
    print(mode, notes)
    assert len(notes) == world


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "local":
    _local_leg()
