"""world_size-2 gloo test of the multi-GPU layer (trajoptkp_amd/distributed.py): sharding covers the
batch exactly once and the single 8-double all-reduce reproduces the single-process reduction."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from trajoptkp_amd import distributed as kd


def test_shard_range_partitions_exactly():
    for n in (1, 7, 64, 1000, 1024):
        for world in (1, 2, 3, 4, 8):
            got = []
            for r in range(world):
                lo, hi = kd.shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                got += list(range(lo, hi))
            assert got == list(range(n))
            sizes = [kd.shard_range(n, r, world)[1] - kd.shard_range(n, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, cost, dJ, status, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = kd.shard_range(cost.shape[0], rank, world)
    v = kd.pack_linesearch(cost[lo:hi], dJ[lo:hi], status[lo:hi])
    kd.allreduce_linesearch(v)
    out_q.put((rank, v.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_linesearch_allreduce_gloo_world2():
    rng = np.random.default_rng(3)
    B = 37
    cost = torch.from_numpy(rng.standard_normal((B, 6)))
    dJ = torch.from_numpy(rng.standard_normal(B))
    status = torch.from_numpy((rng.uniform(size=B) < 0.2).astype(np.int32) * 5)
    ref = kd.pack_linesearch(cost, dJ, status).numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cost, dJ, status, q)) for r in range(2)]
    for p in procs: p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs: p.join(timeout=60)
    for r in range(2):
        assert np.allclose(res[r], ref, rtol=1e-13, atol=1e-13)
    i, alpha, val, n_ok = kd.best_common_alpha(res[0], [(k / 6) ** 2 for k in range(1, 7)])
    assert n_ok == int((status == 0).sum()) and val == res[0][:6].min()
