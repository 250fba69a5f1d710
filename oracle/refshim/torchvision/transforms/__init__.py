from . import functional  # noqa: F401


class ColorJitter(object):
    def __init__(self, *a, **k):
        pass

    def __call__(self, x):
        return x
