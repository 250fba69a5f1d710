class COCO(object):
    pass
