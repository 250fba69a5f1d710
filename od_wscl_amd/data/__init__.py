"""Data boundary of the hot path: transforms, collation, proposal files, datasets,
samplers.  Geometry and bookkeeping on the host like the reference; pixels on the GPU (csrc/preprocess.hip)."""
from .collate_batch import BatchCollator, BBoxAugCollator  # noqa: F401
from .transforms import build_transforms  # noqa: F401
from .build import make_data_loader, build_dataset, DatasetCatalog  # noqa: F401
