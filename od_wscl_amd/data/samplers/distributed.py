"""DistributedSampler (wetectron/data/samplers/distributed.py:10-66): rank r of N sees every N-th slice of an
epoch-seeded permutation padded to a multiple of N -- the data path needs no collective (images are the sharded
unit)."""
import math

import torch
import torch.distributed as dist
from torch.utils.data.sampler import Sampler


class DistributedSampler(Sampler):
    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        if num_replicas is None:
            num_replicas = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.dataset, self.num_replicas, self.rank, self.shuffle = dataset, num_replicas, rank, shuffle
        self.epoch = 0
        self.num_samples = int(math.ceil(len(self.dataset) * 1.0 / self.num_replicas))
        self.total_size = self.num_samples * self.num_replicas

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.epoch)
            indices = torch.randperm(len(self.dataset), generator=g).tolist()
        else:
            indices = torch.arange(len(self.dataset)).tolist()
        indices += indices[: (self.total_size - len(indices))]
        offset = self.num_samples * self.rank
        return iter(indices[offset: offset + self.num_samples])

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch
