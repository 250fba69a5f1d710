class CocoDetection(object):
    def __init__(self, *a, **k):
        raise RuntimeError("dataset stub")
