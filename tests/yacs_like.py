"""cfg for the data-boundary tests, from the values recorded in the golden file."""
from od_wscl_amd.config import make_defaults


def cfg_for_data(golden):
    cfg = make_defaults()
    cfg.INPUT.MIN_SIZE_TRAIN = tuple(int(v) for v in golden["min_train"].tolist())
    cfg.INPUT.MAX_SIZE_TRAIN = int(golden["max_train"])
    cfg.INPUT.MIN_SIZE_TEST = int(golden["min_test"])
    cfg.INPUT.MAX_SIZE_TEST = int(golden["max_test"])
    cfg.INPUT.PIXEL_MEAN = [float(v) for v in golden["pixel_mean"].tolist()]
    cfg.INPUT.PIXEL_STD = [float(v) for v in golden["pixel_std"].tolist()]
    cfg.INPUT.TO_BGR255 = bool(golden["to_bgr255"])
    cfg.INPUT.PCA = bool(golden["pca"])
    return cfg
