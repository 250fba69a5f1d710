import contextlib


def float_function(fn):
    return fn


def half_function(fn):
    return fn


def initialize(model, optimizer=None, opt_level="O0", **kw):
    return (model, optimizer) if optimizer is not None else model


@contextlib.contextmanager
def scale_loss(loss, optimizer, **kw):
    yield loss


def init(*a, **k):
    return None
