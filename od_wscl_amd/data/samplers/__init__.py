from .distributed import DistributedSampler  # noqa: F401
from .grouped_batch_sampler import GroupedBatchSampler  # noqa: F401
from .iteration_based_batch_sampler import IterationBasedBatchSampler  # noqa: F401
