from . import weight_init  # noqa: F401
