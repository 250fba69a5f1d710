class SummaryWriter(object):
    def __init__(self, *a, **k):
        pass
