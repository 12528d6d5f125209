"""kd6d_comm_* (include/kd6d.h): the RCCL communicator behind the C ABI, on the one GPU a test box has.
A one-rank communicator goes through the same librccl entry points as N ranks (ncclCommInitRank,
ncclAllReduce(avg), ncclBroadcast, ncclCommDestroy), stream-ordered on a side stream; the N = 2 semantics of the
Python route that chooses between it and torch.distributed are covered on CPU (tests/test_distributed_cpu.py)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_roundtrip(gpu_device):
    from kd6d._lib import check, lib
    torch.cuda.set_device(gpu_device)
    assert lib.kd6d_comm_version() > 20000
    ident = ctypes.create_string_buffer(128)
    check(lib.kd6d_comm_unique_id(ident), "kd6d_comm_unique_id")
    assert any(b != 0 for b in ident.raw)
    comm = ctypes.c_void_p()
    check(lib.kd6d_comm_init(ctypes.byref(comm), 0, 1, ident.raw), "kd6d_comm_init")
    assert lib.kd6d_comm_rank(comm) == 0 and lib.kd6d_comm_world(comm) == 1
    side = torch.cuda.Stream()
    g = torch.arange(2_300_000, dtype=torch.float32, device=gpu_device) * 1e-3     # the tiny-H gradient bucket's size
    want = g.clone()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        st = ctypes.c_void_p(side.cuda_stream)
        g.mul_(2.0)                                                               # stream order: runs before ...
        check(lib.kd6d_comm_allreduce(comm, ctypes.c_void_p(g.data_ptr()), g.numel(), 1, st), "allreduce mean")
        check(lib.kd6d_comm_allreduce(comm, ctypes.c_void_p(g.data_ptr()), g.numel(), 0, st), "allreduce sum")
        check(lib.kd6d_comm_broadcast(comm, ctypes.c_void_p(g.data_ptr()), g.numel() * 4, 0, st), "broadcast")
        g.mul_(0.5)                                                               # ... and this one after
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(g, want)
    assert lib.kd6d_comm_broadcast(comm, ctypes.c_void_p(g.data_ptr()), 4, 3, None) == -1      # root outside the world
    assert b"bad arguments" in lib.kd6d_last_error()
    check(lib.kd6d_comm_destroy(comm), "kd6d_comm_destroy")
    assert lib.kd6d_comm_allreduce(None, None, 0, 1, None) == -1


def test_overlapped_exchange_captured_in_the_step_graph(gpu_device):
    """EXCHANGE_MODE "overlap" (kd6d/libs/distributed.py): the FPN + head slice of the gradient bucket is all-reduced on a
    side stream beside the backbone sweep, the backbone's slice behind it, both collectives CAPTURED inside the single
    step graph.  On the one GPU of a test box: a one-rank process group + communicator (the rehearsal switch of
    bench.py --rccl-single-rank), three replayed steps -- the mean over one rank is the identity, so losses, gradient
    norm and parameters must equal those of the same steps without any exchange."""
    import os
    import socket
    import torch.distributed as dist
    from kd6d.graph import GraphedKDStep
    from kd6d.kd_losses import PackedTargets
    from kd6d.libs import distributed as D
    from kd6d.libs.poses import ImageList
    from kd6d.optim import FusedClipAdamW
    from kd6d.synthetic import make_batch
    from test_step_gpu import build
    dev = gpu_device
    torch.cuda.set_device(dev)
    bias = [1.0] + [-6.0] * 14
    B, crop = 4, 128
    batches = []
    for i in range(2):
        images, targets = make_batch(B, 7 + i, crop=crop)
        batches.append((ImageList(images.tensors.to(dev), images.sizes), PackedTargets(targets, dev)))
    keys = torch.rand(B * 340, generator=torch.Generator().manual_seed(3)).to(dev)

    def run(exchange):
        teacher = build("darknet53", "bf16", 2, dev, bias).eval()
        student = build("darknet_tiny_h", "bf16", 1, dev).train()
        student._debug_keys = keys
        opt = FusedClipAdamW(student, lr=1e-3)
        gs = GraphedKDStep(teacher, student, opt, (0.1, 1.0, 5.0), pipeline=False, exchange=exchange)
        out = []
        for i in range(3):
            ld = gs(*batches[i % 2])
            torch.cuda.synchronize()
            out.append([float(ld[k]) for k in ("loss_cls", "loss_reg", "loss_kd")] + [float(opt.grad_norm())])
        return out, student.net.store.params.detach().clone(), gs.graphs_per_step

    base, p_base, n_base = run(None)
    assert n_base == 1
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group(backend="nccl", init_method="env://")
    try:
        D.SINGLE_RANK_EXCHANGE = True
        assert D.init_exchange().startswith("kd6d_comm")
        got, p_got, n_got = run("overlap")
        assert n_got == 1                                   # the collectives are inside the one graph
        got2, p_got2, n_got2 = run("between")
        assert n_got2 == 2
    finally:
        D.SINGLE_RANK_EXCHANGE = False
        D.shutdown_exchange()
        dist.destroy_process_group()
    # bf16 steps repeat to atomic-order noise on the first step; AdamW's lr * sign(g) turns that into a few percent of
    # the losses two updates later (B = 4 at 128 x 128: measured 3 %)
    for other in (got, got2):
        assert base[0] == pytest.approx(other[0], rel=1e-2), (base, other)
        for a, b in zip(base[1:], other[1:]):
            assert a == pytest.approx(b, rel=0.15), (base, other)
    for q in (p_got, p_got2):
        # three AdamW steps of lr 1e-3 move a parameter by <= 3e-3; where the runs disagree about the sign of a near-zero
        # gradient the two copies end up to 6e-3 apart -- a few elements; the bulk moves together
        d = (q - p_base).abs()
        assert float(d.max()) <= 6.5e-3 and float(d.mean()) <= 8e-4, (float(d.max()), float(d.mean()))
