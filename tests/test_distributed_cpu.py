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
    # the exchange step GraphedKDStep._exchange / PoseModuleKD._run_backward drive: the trainable slice of the flat
    # gradient bucket of a real parameter store; registered-but-unused parameters stay out of the collective
    from kd6d import engine
    st = engine.PoseNet("darknet_tiny_h", torch.float32).store
    st.grads = torch.full((st.params.numel(),), float(rank + 1))       # oversize on purpose: [n_train:] must not move
    assert D.exchange_route() in ("none", "torch.distributed (gloo)")
    D.init_exchange()
    assert D.exchange_route() == "torch.distributed (gloo)"
    D.exchange_gradients(st)
    frozen = sorted(e.name for e in st.order if e.region == "frozen")
    # the two-slice schedule (EXCHANGE_MODE "overlap": FPN + head first, the backbone's slice afterwards) moves exactly
    # the same elements: every trainable gradient once, nothing behind n_train
    split = D.bucket_split(st)
    two = torch.full((st.params.numel(),), float(rank + 1))
    one = st.grads
    st.grads = two
    D.exchange_slice(st, split, st.n_train)
    head_done = two.clone()
    D.exchange_slice(st, 0, split)
    st.grads = one
    names_before = [e.name for e in st.order if e.region == "train" and e.offset < split]
    out[rank] = (g.clone(), p.clone(), st.grads.clone(), st.n_train, frozen, two.clone(), head_done, split, names_before)
    dist.destroy_process_group()


def test_allreduce_mean_and_broadcast_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    exp = torch.arange(1000, dtype=torch.float32) * 1.5
    for r in range(world):
        g, p, bucket, n_train, frozen, two, head_done, split, names_before = out[r]
        assert torch.equal(g, exp)
        assert torch.equal(p, torch.zeros(64))
        assert 2_200_000 < n_train < bucket.numel()
        assert torch.equal(bucket[:n_train], torch.full((n_train,), 1.5))              # mean of rank gradients
        assert torch.equal(bucket[n_train:], torch.full((bucket.numel() - n_train,), float(r + 1)))
        assert frozen == ["backbone.output.final_conv.bias", "backbone.output.final_conv.weight", "head.scales.4.scale"]
        assert torch.equal(two, bucket)                                               # two slices == one all-reduce
        assert 0 < split < 0.15 * n_train                                             # the backbone is the small slice
        assert names_before and all(n.startswith("backbone.") for n in names_before)
        assert torch.equal(head_done[split:n_train], torch.full((n_train - split,), 1.5))
        assert torch.equal(head_done[:split], torch.full((split,), float(r + 1)))    # ... not exchanged yet at that point


def test_single_process_helpers():
    import sys
    from kd6d.libs import distributed as D
    assert D.get_world_size() == 1 and D.get_rank() == 0
    t = torch.ones(4)
    assert D.allreduce_mean_(t) is t and D.shard_batch(16) == 16
