"""Stand-in for torchvision (absent here).  Only `ops.nms` does arithmetic; it
restates torchvision's documented semantics: IoU without the +1 pixel
convention, a box is suppressed when IoU > threshold (strict), kept indices are
returned in descending-score order."""
from . import ops, models, transforms, datasets  # noqa: F401
