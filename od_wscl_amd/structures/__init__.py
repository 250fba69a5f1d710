from .bounding_box import BoxList
from .image_list import ImageList, to_image_list
from .boxlist_ops import boxlist_iou, boxlist_nms_index, cat_boxlist

__all__ = ["BoxList", "ImageList", "to_image_list", "boxlist_iou", "boxlist_nms_index", "cat_boxlist"]
