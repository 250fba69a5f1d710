"""IterationBasedBatchSampler (wetectron/data/samplers/iteration_based_batch_sampler.py:5-31): re-iterates a batch
sampler until `num_iterations` batches were produced; the iteration count is the epoch seed of a DistributedSampler."""
from torch.utils.data.sampler import BatchSampler


class IterationBasedBatchSampler(BatchSampler):
    def __init__(self, batch_sampler, num_iterations, start_iter=0):
        self.batch_sampler = batch_sampler
        self.num_iterations = num_iterations
        self.start_iter = start_iter

    def __iter__(self):
        iteration = self.start_iter
        while iteration <= self.num_iterations:
            if hasattr(self.batch_sampler.sampler, "set_epoch"):
                self.batch_sampler.sampler.set_epoch(iteration)
            for batch in self.batch_sampler:
                iteration += 1
                if iteration > self.num_iterations:
                    break
                yield batch

    def __len__(self):
        return self.num_iterations
