"""Rank sharding of the image list (the reference's wetectron/data/samplers/distributed.py:42-60 semantics, pinned by
tests/golden/sampler_voc.npz): images are the data-parallel unit, so the data path needs no
collective -- every rank derives the same epoch order from the epoch number and takes its own CONTIGUOUS block.
(torch.utils.data.DistributedSampler strides instead of blocking, i.e. it would hand different images to a rank.)"""
import torch
import torch.distributed as dist
from torch.utils.data.sampler import Sampler


def epoch_order(n, epoch, shuffle):
    """The order all ranks agree on for one epoch: a permutation seeded by the epoch number, or 0..n-1."""
    if not shuffle:
        return list(range(n))
    gen = torch.Generator()
    gen.manual_seed(epoch)
    return torch.randperm(n, generator=gen).tolist()


def rank_block(order, world, rank):
    """Block `rank` of `world` equal blocks; the list wraps around so that every rank gets ceil(n / world) items."""
    per_rank = -(-len(order) // world)
    wrapped = order + order[:per_rank * world - len(order)]
    return wrapped[rank * per_rank:(rank + 1) * per_rank]


def _dist_default(value, getter, fallback):
    if value is not None:
        return value
    return getter() if dist.is_available() and dist.is_initialized() else fallback


class DistributedSampler(Sampler):
    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        self.dataset = dataset
        self.num_replicas = _dist_default(num_replicas, dist.get_world_size, 1)
        self.rank = _dist_default(rank, dist.get_rank, 0)
        self.shuffle = shuffle
        self.epoch = 0

    @property
    def num_samples(self):
        return -(-len(self.dataset) // self.num_replicas)

    @property
    def total_size(self):
        return self.num_samples * self.num_replicas

    def __iter__(self):
        return iter(rank_block(epoch_order(len(self.dataset), self.epoch, self.shuffle), self.num_replicas, self.rank))

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch
