"""world_size-2 gloo tests (CPU) of the N>1 host path: ring partition of the key space is identical on
every rank, disjoint and complete; the max-over-ranks timing reduction; a sharded evaluation in which
every rank evaluates only the keys it owns reproduces a single unsharded oracle bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import streams
    import support
    from gubernator_amd import shard
    K = 60_000
    table = streams.key_table(K)
    mine = shard.owned_key_ids(table, world, rank, chunk=25_000)
    # partition is disjoint and complete
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([len(mine)], dtype=torch.int64))
    total = sum(int(c.item()) for c in counts)
    sums = torch.tensor([int(mine.sum())], dtype=torch.int64)
    dist.all_reduce(sums)
    assert total == K, (total, K)
    assert int(sums.item()) == K * (K - 1) // 2
    assert shard.sum_over_ranks(len(mine)) == K
    # every rank derives the same ownership for any key (ring is deterministic)
    probe = np.arange(0, K, 97)
    owner_here = np.isin(probe, mine).astype(np.int64)
    allown = [torch.zeros(len(probe), dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allown, torch.from_numpy(owner_here))
    assert (sum(allown) == 1).all()
    # timing contract: max over ranks
    assert shard.max_over_ranks(1.0 + rank) == float(world)
    # sharded evaluation == unsharded oracle: rank-local oracle over owned keys vs one global oracle
    rng = np.random.default_rng(1234)       # same stream on every rank
    glob = support.Oracle(cache_size=1 << 20)
    local = support.Oracle(cache_size=1 << 20)
    now = streams.NOW0
    for step in range(5):
        ids = rng.zipf(1.2, 4000) % K
        want = glob.eval(streams.bench_batch(table, ids, now + step * 7000, algorithm=step % 2, limit=5))
        sel = np.nonzero(np.isin(ids, mine))[0]
        got = local.eval(streams.bench_batch(table, ids[sel], now + step * 7000, algorithm=step % 2, limit=5))
        for name in ("status", "remaining", "reset_time", "limit"):
            assert np.array_equal(getattr(got, name)[:len(sel)], getattr(want, name)[:len(ids)][sel]), (step, name)
    dist.barrier()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


def test_two_rank_partition_and_sharded_eval(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_ring_balance_8_gpus():
    """config 4 shape: 8 peers gpu0..gpu7, 512 vnodes each: every shard within +-25% of the mean."""
    sys.path.insert(0, os.path.dirname(HERE))
    import streams
    from gubernator_amd import shard
    table = streams.key_table(400_000)
    sizes = [len(shard.owned_key_ids(table, 8, r)) for r in range(8)]
    assert sum(sizes) == 400_000
    assert max(sizes) < 1.25 * 50_000 and min(sizes) > 0.75 * 50_000, sizes
