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


# ---------------------------------------------------------------------------------------------------------------------
# The control flow of train_kd.py's data-parallel start in the grouped launch mode ("graphs first": the step's hipGraphs
# are recorded before the first communicator exists) and its collective stop, at world size 2 on gloo with the device
# work replaced by a ledger -- the path the first multi-GPU box would otherwise be the first to execute
# (reference: libs/train_libs.py:117,123-130,272 -- lr / N, the DDP constructor's broadcast, batch / N; train_kd.py:43-51).
# ---------------------------------------------------------------------------------------------------------------------
class _Store:
    def __init__(self, rank, n):
        self.params = torch.full((n,), float(rank + 1))
        self.bufs = torch.full((8,), float(10 * (rank + 1)))


class _Net:
    def __init__(self, name, rank, n, ledger):
        self.store, self.name, self.ledger = _Store(rank, n), name, ledger

    def refresh_derived_in_place(self, need_dgrad=False):
        # what the recorded kernels read must be recomputed from the BROADCAST weights: rank 0's values are in place now
        self.ledger.append(("refresh", self.name, bool(need_dgrad), float(self.store.params[0]), float(self.store.bufs[0])))


class _Model:
    def __init__(self, name, rank, n, ledger):
        self.net = _Net(name, rank, n, ledger)


class _GStep:
    def __init__(self, ledger):
        self.ledger = ledger

    def prepare(self, images, targets):
        from kd6d.libs import distributed as D
        # no communicator and no collective may exist yet when the graphs are recorded
        self.ledger.append(("prepare", images, targets, D.exchange_route()))


def _dp_worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "kd-6d-pose-adlp_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from kd6d.libs import distributed as D
    from kd6d.libs import train_libs as T
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ledger, printed = [], []
    teacher, student = _Model("teacher", rank, 100, ledger), _Model("student", rank, 60, ledger)
    route = T.start_exchange_after_graphs(_GStep(ledger), teacher, student, ("images0", "targets0", None), log=printed.append)
    ledger.append(("route", route, D.exchange_route()))
    # a step's exchange after the start: the route that was set up carries it
    g = torch.full((32,), float(rank + 1))
    D.allreduce_mean_(g)
    # the collective stop: only rank 1 saw a barrier give up -- BOTH ranks must stop, and leave the group cleanly
    stopped = None
    try:
        T.stop_if_barrier_timeouts(0, True)                      # nobody timed out: training goes on
        T.stop_if_barrier_timeouts(3 if rank == 1 else 0, True)
    except SystemExit as e:
        stopped = str(e)
    out[rank] = (ledger, printed, teacher.net.store.params.clone(), student.net.store.bufs.clone(), g, stopped,
                 dist.is_initialized(), D.exchange_route())


def test_graphs_first_start_and_collective_stop_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        ledger, printed, t_params, s_bufs, g, stopped, still_init, route_after = out[r]
        kinds = [e[0] for e in ledger]
        assert kinds == ["prepare", "refresh", "refresh", "route"], kinds
        assert ledger[0] == ("prepare", "images0", "targets0", "none")       # graphs recorded before any exchange exists
        # both refreshes ran AFTER the broadcast: they saw rank 0's parameters (1.0) and buffers (10.0) on every rank
        assert ledger[1] == ("refresh", "teacher", False, 1.0, 10.0)
        assert ledger[2] == ("refresh", "student", True, 1.0, 10.0)
        assert ledger[3][1] == ledger[3][2] == "torch.distributed (gloo)"
        assert printed == ["gradient exchange: torch.distributed (gloo)"]
        assert torch.equal(t_params, torch.ones(100)) and torch.equal(s_bufs, torch.full((8,), 10.0))
        assert torch.equal(g, torch.full((32,), 1.5))
        assert stopped is not None and "3 in-kernel barrier waits" in stopped       # rank 0 stops although ITS counter was 0
        assert not still_init and route_after == "none"

