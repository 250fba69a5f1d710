from . import coco  # noqa: F401
