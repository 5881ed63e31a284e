"""world_size-2 gloo test of the batch-sharded forward: gathered logits == single-process logits.
The per-rank compute here is the CPU oracle (allowed in tests); the product's sharding logic
(mgnns_amd/sharded.py) is what is under test."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mgnns_amd import synth
from mgnns_amd.sharded import ShardedForward, gather_variable, shard_bounds


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_forward(x):            # per-sample function: no cross-sample coupling, like the model in eval
    return torch.stack([x.sum(1), (x * x).sum(1), x[:, 0]], 1)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rs = np.random.RandomState(0)
    x = torch.from_numpy(rs.standard_normal((10, 5)).astype(np.float32))
    # equal shards through ShardedForward
    lo, hi = shard_bounds(10, world, rank)
    out = ShardedForward(_fake_forward)(x[lo:hi])
    ok1 = torch.equal(out, _fake_forward(x))
    # unequal shards (7 samples over 2 ranks) through gather_variable
    lo, hi = shard_bounds(7, world, rank)
    out2 = gather_variable(_fake_forward(x[:7][lo:hi]))
    ok2 = torch.equal(out2, _fake_forward(x[:7]))
    q.put((rank, bool(ok1), bool(ok2), tuple(out.shape)))
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 256, 257):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_two_rank_gloo_gather_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok1, ok2, shape in res:
        assert ok1 and ok2 and shape == (10, 3)
