class COCOeval(object):
    pass
