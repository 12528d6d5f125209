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
