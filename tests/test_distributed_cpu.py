"""CPU, world_size 2, gloo: the one exchange step of the data-parallel path (mean all-reduce of the
flat gradient bucket, parameter broadcast, per-rank batch sharding)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "kd-6d-pose-adlp_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from kd6d.libs import distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert D.get_world_size() == world and D.get_rank() == rank
    assert D.shard_batch(16) == 8
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)        # rank-dependent "gradients"
    D.allreduce_mean_(g)
    p = torch.full((64,), float(rank))
    D.broadcast_(p, 0)
    D.synchronize()
    out[rank] = (g.clone(), p.clone())
    dist.destroy_process_group()


def test_allreduce_mean_and_broadcast_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    exp = torch.arange(1000, dtype=torch.float32) * 1.5
    for r in range(world):
        g, p = out[r]
        assert torch.equal(g, exp)
        assert torch.equal(p, torch.zeros(64))


def test_single_process_helpers():
    import sys
    from kd6d.libs import distributed as D
    assert D.get_world_size() == 1 and D.get_rank() == 0
    t = torch.ones(4)
    assert D.allreduce_mean_(t) is t and D.shard_batch(16) == 16
