"""Process-group helpers with the reference's names (libs/distributed.py:9-41) plus the
gradient exchange the reference lacks (its DDP wrapper is discarded, libs/train_libs.py:124-130).

MI355X: one process per GPU, backend "nccl" (= RCCL over xGMI).  The data path has exactly one
collective per step: a mean all-reduce of the single flat fp32 gradient bucket (2.3 M / 8.5 M
elements) -- latency-bound at these sizes, so one bucket, no overlap machinery.
"""
import math
import os

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


def exchange_active():
    """True when the per-step gradient exchange has to run: more than one rank, or a one-rank process group with
    KD6D_EXCHANGE_SINGLE_RANK=1 (a rehearsal of the RCCL path -- communicator set-up, the collective between the
    two replayed graphs, the barriers -- on a box with one GPU)."""
    if not dist.is_available() or not dist.is_initialized():
        return False
    return dist.get_world_size() > 1 or os.environ.get("KD6D_EXCHANGE_SINGLE_RANK") == "1"


def allreduce_mean_(flat):
    """In-place mean over ranks of one flat bucket."""
    n = get_world_size()
    if not exchange_active():
        return flat
    if dist.get_backend() == "nccl":
        dist.all_reduce(flat, op=dist.ReduceOp.AVG)
    else:  # gloo (CPU tests): no AVG
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(n)
    return flat


def broadcast_(flat, src=0):
    if get_world_size() > 1:
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
