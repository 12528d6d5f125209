"""Process-group helpers with the reference's names (libs/distributed.py:9-41) plus the
gradient exchange the reference lacks (its DDP wrapper is discarded, libs/train_libs.py:124-130).

MI355X: one process per GPU.  The data path has exactly one collective per step: a mean all-reduce
of the single flat fp32 gradient bucket (2.3 M / 8.5 M elements) -- latency-bound at these sizes, so
one bucket, no overlap machinery.  The collective goes through the C ABI (kd6d_comm_*: a communicator
this library owns on librccl, include/kd6d.h), enqueued on the step's HIP stream between its two
replayed graphs; torch.distributed is only the rendezvous (the 128-byte RCCL id travels over its
store) and the fall-back route for CPU tensors (the gloo tests) or when librccl cannot be opened.
"""
import ctypes
import math

import torch
from torch import distributed as dist
from torch.utils.data import sampler


def get_rank():
    if not dist.is_available() or not dist.is_initialized():
        return 0
    return dist.get_rank()


def get_world_size():
    if not dist.is_available() or not dist.is_initialized():
        return 1
    return dist.get_world_size()


def synchronize():
    if get_world_size() > 1:
        dist.barrier()


# bench.py --rccl-single-rank sets this: a one-rank process group still runs the exchange (a rehearsal of the RCCL
# path -- communicator set-up, the collective between the two replayed graphs, the barriers -- on a box with one GPU)
SINGLE_RANK_EXCHANGE = False


def exchange_active():
    """True when the per-step gradient exchange has to run: more than one rank, or SINGLE_RANK_EXCHANGE."""
    if not dist.is_available() or not dist.is_initialized():
        return False
    return dist.get_world_size() > 1 or SINGLE_RANK_EXCHANGE


_comm = None            # kd6d_comm* of this process (ctypes.c_void_p) once init_exchange() succeeded
_route = "none"


def exchange_route():
    """Which library carries the per-step collective: 'kd6d_comm (librccl x.y.z)', 'torch.distributed (<backend>)'
    or 'none' (single process)."""
    return _route


def init_exchange():
    """Collective (every rank calls it once, after init_process_group and torch.cuda.set_device): build this
    process's kd6d communicator.  Rank 0 creates the RCCL id, the process group's store hands it out."""
    global _comm, _route
    if not exchange_active():
        _route = "none"
        return _route
    _route = "torch.distributed (%s)" % dist.get_backend()
    if _comm is not None or not torch.cuda.is_available():
        return _route
    from .._lib import lib
    # Agree BEFORE the blocking collective: ncclCommInitRank waits for every rank, so a rank that cannot take part
    # (no librccl, no current device) must be known to all of them first -- it would otherwise fall through to the
    # torch route while the others wait for it inside the init.
    able = 1 if lib.kd6d_comm_version() >= 0 else 0          # dlopen + symbol resolution, no communication
    try:
        torch.cuda.current_device()
    except (RuntimeError, AssertionError):
        able = 0
    if not _all_agree(able):
        return _route            # some rank has no librccl / device: every rank stays on torch.distributed
    ident = ctypes.create_string_buffer(128)
    ok = 1
    if get_rank() == 0 and lib.kd6d_comm_unique_id(ident) != 0:
        ok = 0
    box = [ident.raw if ok else None]
    dist.broadcast_object_list(box, src=0)
    if box[0] is None:
        return _route            # rank 0 could not create the id: every rank stays on torch.distributed
    handle = ctypes.c_void_p()
    rc = lib.kd6d_comm_init(ctypes.byref(handle), get_rank(), dist.get_world_size(), box[0])
    if _all_agree(1 if rc == 0 else 0):
        _comm = handle
        v = lib.kd6d_comm_version()
        _route = "kd6d_comm (librccl %d.%d.%d)" % (v // 10000, (v // 100) % 100, v % 100)
    elif rc == 0:
        lib.kd6d_comm_destroy(handle)
    return _route


def _all_agree(flag):
    """MIN over ranks of a 0 / 1 flag through torch.distributed (the rendezvous route, always available here)."""
    t = torch.tensor([int(flag)], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item()) == 1


def max_over_ranks(value):
    """MAX over ranks of a small non-negative integer (e.g. the barrier-timeout count): every rank gets the same
    answer, so a decision taken on it is collective."""
    if get_world_size() <= 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def shutdown_exchange():
    global _comm, _route
    if _comm is not None:
        from .._lib import lib
        torch.cuda.synchronize()
        lib.kd6d_comm_destroy(_comm)
        _comm = None
    _route = "none"


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def allreduce_mean_(flat):
    """In-place mean over ranks of one flat bucket (asynchronous on the current HIP stream)."""
    n = get_world_size()
    if not exchange_active():
        return flat
    if _comm is not None and flat.is_cuda:
        from .._lib import check, lib
        assert flat.dtype == torch.float32 and flat.is_contiguous()
        check(lib.kd6d_comm_allreduce(_comm, ctypes.c_void_p(flat.data_ptr()), flat.numel(), 1, _stream()),
              "kd6d_comm_allreduce")
        return flat
    if dist.get_backend() == "nccl":
        dist.all_reduce(flat, op=dist.ReduceOp.AVG)
    else:  # gloo (CPU tests): no AVG
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(n)
    return flat


# How the per-step exchange is scheduled (bench.py --exchange, train_kd.py --exchange):
#   "between"  one mean all-reduce of the whole trainable slice, issued eagerly BETWEEN the step's two replayed graphs
#              (backward | optimiser).  The default: every piece of it has run on hardware at world size 1 and its
#              arithmetic at world size 2 (gloo); nothing about it depends on capturing a collective.
#   "overlap"  two slices.  The FPN + head gradients (90 % of the tiny-H bucket) are final when the reverse sweep leaves
#              the FPN, 0.6 ms before the step ends: their all-reduce goes out there on a side stream and runs beside the
#              backbone sweep; the backbone's small slice follows when the sweep has joined.  GraphedKDStep captures
#              both collectives INSIDE the step's single graph (RCCL supports stream capture), so no launch gap and no
#              host call is left on the step's critical path.  Validated at world size 1 on hardware (rehearsal) and at
#              world size 2 on CPU (eager, gloo); opt-in until an N > 1 box has replayed a captured collective.
EXCHANGE_MODE = "between"


def bucket_split(store):
    """First element of the FPN + head part of the flat gradient bucket (the backbone's gradients lie before it: the
    parameter store registers backbone, FPN, head in that order)."""
    first = None
    for e in store.order:
        if e.region != "train":
            continue
        if e.name.startswith("backbone."):
            assert first is None, "backbone entry %s behind a non-backbone entry" % e.name
        elif first is None:
            first = e.offset
    return store.n_train if first is None else first


def exchange_slice(store, lo, hi):
    """Mean over ranks of grads[lo:hi] (asynchronous on the current stream)."""
    if exchange_active() and hi > lo:
        allreduce_mean_(store.grads[lo:hi])


def exchange_gradients(store):
    """THE exchange step of the data-parallel path: mean over ranks of the trainable slice of the flat gradient
    bucket.  Parameters the reference registers but never gives a gradient (backbone.output.*, head.scales.4 of a
    4-level student) live behind n_train and stay out of the collective."""
    if exchange_active():
        allreduce_mean_(store.grads[:store.n_train])


def broadcast_(flat, src=0):
    if _comm is not None and flat.is_cuda and exchange_active():
        from .._lib import check, lib
        assert flat.is_contiguous()
        check(lib.kd6d_comm_broadcast(_comm, ctypes.c_void_p(flat.data_ptr()), flat.numel() * flat.element_size(),
                                      src, _stream()), "kd6d_comm_broadcast")
    elif get_world_size() > 1:
        dist.broadcast(flat, src)
    return flat


def shard_batch(global_batch):
    """Per-rank batch: IMS_PER_BATCH / N_GPU (libs/train_libs.py:272)."""
    n = get_world_size()
    if global_batch % n:
        raise ValueError("global batch %d is not divisible by %d ranks" % (global_batch, n))
    return global_batch // n


class DistributedSampler(sampler.Sampler):
    """Epoch-seeded permutation, wrap-around padding, contiguous rank slice
    (semantics of libs/distributed.py:109-165)."""

    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        self.dataset = dataset
        self.num_replicas = get_world_size() if num_replicas is None else num_replicas
        self.rank = get_rank() if rank is None else rank
        self.epoch = 0
        self.num_samples = int(math.ceil(len(dataset) * 1.0 / self.num_replicas))
        self.total_size = self.num_samples * self.num_replicas
        self.shuffle = shuffle

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.epoch)
            indices = torch.randperm(len(self.dataset), generator=g).tolist()
        else:
            indices = list(range(len(self.dataset)))
        indices += indices[: (self.total_size - len(indices))]
        offset = self.num_samples * self.rank
        return iter(indices[offset:offset + self.num_samples])

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch
