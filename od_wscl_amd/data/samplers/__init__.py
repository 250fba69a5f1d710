from .distributed import DistributedSampler  # noqa: F401
from .grouped_batch_sampler import GroupedBatchSampler  # noqa: F401
from .repeating import IterationBasedBatchSampler  # noqa: F401
