"""A batch sampler stretched to a fixed number of training iterations.

Behaviour of the reference's IterationBasedBatchSampler (wetectron/data/samplers/iteration_based_batch_sampler.py:5-31),
which the trainer's iteration-counted loop relies on (engine/trainer.py:79-100): the wrapped batch sampler is walked
again and again until `num_iterations` batches exist in total; a resumed run starts counting at `start_iter`; and at the
start of every pass the number of batches produced so far is handed to the underlying sampler as its shuffling epoch
(DistributedSampler.set_epoch), so that every pass -- on every rank alike -- draws a new permutation."""
from itertools import islice


class IterationBasedBatchSampler(object):
    def __init__(self, batch_sampler, num_iterations, start_iter=0):
        self.batch_sampler = batch_sampler
        self.num_iterations = int(num_iterations)
        self.start_iter = int(start_iter)

    def __len__(self):
        return self.num_iterations

    def __iter__(self):
        done = self.start_iter
        reseed = getattr(getattr(self.batch_sampler, "sampler", None), "set_epoch", None)
        while done < self.num_iterations:
            if reseed is not None:
                reseed(done)
            before = done
            for batch in islice(iter(self.batch_sampler), self.num_iterations - done):
                done += 1
                yield batch
            if done == before:
                raise RuntimeError("IterationBasedBatchSampler: the wrapped batch sampler produced no batch")
