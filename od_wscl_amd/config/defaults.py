"""Defaults of every key the hot path reads, with the reference's names and values
(wetectron/config/defaults.py; line numbers cited per group)."""
from .node import CfgNode as CN


def make_defaults():
    c = CN()
    c.MODEL = CN()
    c.MODEL.DEVICE = "cuda"                                   # :27
    c.MODEL.META_ARCHITECTURE = "GeneralizedRCNN"             # :28
    c.MODEL.WSOD_ON = False
    c.MODEL.FASTER_RCNN = True
    c.MODEL.CLS_AGNOSTIC_BBOX_REG = False
    c.MODEL.WEIGHT = ""
    c.MODEL.BACKBONE = CN()
    c.MODEL.BACKBONE.CONV_BODY = "R-50-C4"
    c.MODEL.BACKBONE.FREEZE_CONV_BODY_AT = 2                  # :128
    c.MODEL.RESNETS = CN()                                    # :304-330
    c.MODEL.RESNETS.NUM_GROUPS = 1
    c.MODEL.RESNETS.WIDTH_PER_GROUP = 64
    c.MODEL.RESNETS.STRIDE_IN_1X1 = True
    c.MODEL.RESNETS.TRANS_FUNC = "BottleneckWithFixedBatchNorm"
    c.MODEL.RESNETS.STEM_FUNC = "StemWithFixedBatchNorm"
    c.MODEL.RESNETS.RES5_DILATION = 1
    c.MODEL.RESNETS.BACKBONE_OUT_CHANNELS = 256 * 4
    c.MODEL.RESNETS.RES2_OUT_CHANNELS = 256
    c.MODEL.RESNETS.STEM_OUT_CHANNELS = 64
    c.MODEL.RESNETS.STAGE_WITH_DCN = (False, False, False, False)
    c.MODEL.RESNETS.WITH_MODULATED_DCN = False
    c.MODEL.RESNETS.DEFORMABLE_GROUPS = 1
    c.MODEL.ROI_HEADS = CN()
    c.MODEL.ROI_HEADS.FG_IOU_THRESHOLD = 0.5
    c.MODEL.ROI_HEADS.BBOX_REG_WEIGHTS = (10.0, 10.0, 5.0, 5.0)   # :213
    c.MODEL.ROI_HEADS.SCORE_THRESH = 0.05
    c.MODEL.ROI_HEADS.NMS = 0.5
    c.MODEL.ROI_HEADS.DETECTIONS_PER_IMG = 100
    c.MODEL.ROI_BOX_HEAD = CN()
    c.MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR = "ResNet50Conv5ROIFeatureExtractor"
    c.MODEL.ROI_BOX_HEAD.POOLER_METHOD = "ROIAlign"
    c.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION = 14
    c.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO = 0
    c.MODEL.ROI_BOX_HEAD.POOLER_SCALES = (1.0 / 16,)
    c.MODEL.ROI_BOX_HEAD.NUM_CLASSES = 81                     # :243
    c.MODEL.ROI_WEAK_HEAD = CN()
    c.MODEL.ROI_WEAK_HEAD.PREDICTOR = "MISTPredictor"
    c.MODEL.ROI_WEAK_HEAD.LOSS = "RoIRegLoss"
    c.MODEL.ROI_WEAK_HEAD.OICR_P = 0.0
    c.MODEL.ROI_WEAK_HEAD.REGRESS_ON = False
    c.MODEL.ROI_WEAK_HEAD.REGRESS_HEUR = "AVG"
    c.MODEL.ROI_WEAK_HEAD.PARTIAL_LABELS = "none"             # :292
    c.MODEL.ROI_WEAK_HEAD.ROI_LOSS_REFINE = False
    c.DB = CN()
    c.DB.METHOD = "none"
    c.INPUT = CN()
    c.INPUT.MIN_SIZE_TRAIN = (800,)                           # :58-78
    c.INPUT.MAX_SIZE_TRAIN = 1333
    c.INPUT.MIN_SIZE_TEST = 800
    c.INPUT.MAX_SIZE_TEST = 1333
    c.INPUT.PIXEL_MEAN = [102.9801, 115.9465, 122.7717]
    c.INPUT.PIXEL_STD = [1., 1., 1.]
    c.INPUT.TO_BGR255 = True
    c.INPUT.BRIGHTNESS = 0.0
    c.INPUT.CONTRAST = 0.0
    c.INPUT.SATURATION = 0.0
    c.INPUT.HUE = 0.0
    c.INPUT.PCA = True
    c.INPUT.VERTICAL_FLIP_PROB_TRAIN = 0.0
    c.DATASETS = CN()                                         # :83-98
    c.DATASETS.TRAIN = ()
    c.DATASETS.TEST = ()
    c.PROPOSAL_FILES = CN()
    c.PROPOSAL_FILES.TRAIN = ()
    c.PROPOSAL_FILES.TEST = ()
    c.DATALOADER = CN()                                       # :104-113
    c.DATALOADER.NUM_WORKERS = 4
    c.DATALOADER.SIZE_DIVISIBILITY = 0
    c.DATALOADER.ASPECT_RATIO_GROUPING = True
    c.SOLVER = CN()
    c.SOLVER.MAX_ITER = 40000
    c.SOLVER.BASE_LR = 0.001
    c.SOLVER.BIAS_LR_FACTOR = 2
    c.SOLVER.MOMENTUM = 0.9
    c.SOLVER.WEIGHT_DECAY = 0.0005
    c.SOLVER.WEIGHT_DECAY_BIAS = 0
    c.SOLVER.IMS_PER_BATCH = 16
    c.SOLVER.CLASS_BATCH = False                              # :463
    c.SOLVER.CHECKPOINT_PERIOD = 2500                         # :454
    c.SOLVER.GAMMA = 0.1                                      # :445-450
    c.SOLVER.STEPS = (30000,)
    c.SOLVER.WARMUP_FACTOR = 1.0 / 3
    c.SOLVER.WARMUP_ITERS = 500
    c.SOLVER.WARMUP_METHOD = "linear"
    c.SOLVER.CONTRA = False
    c.SOLVER.ITER_SIZE = -1                                   # :459-461 gradient accumulation over ITER_SIZE iterations
    # OD-WSCL hyper-parameters, lower-case top-level keys (:540-551)
    c.nms = 0.1
    c.lmda = 0.1
    c.pos_update = 0
    c.thres = 0.5
    c.iou = 0.5          # unused by the loss (Q9)
    c.temp = 0.2
    c.loss = "supconv2"
    c.TEST = CN()
    c.TEST.BBOX_AUG = CN()
    c.TEST.IMS_PER_BATCH = 8                                  # :504
    c.TEST.BBOX_AUG.ENABLED = False                           # :512-531
    c.TEST.BBOX_AUG.HEUR = "UNION"
    c.TEST.BBOX_AUG.H_FLIP = False
    c.TEST.BBOX_AUG.SCALES = ()
    c.TEST.BBOX_AUG.MAX_SIZE = 4000
    c.TEST.BBOX_AUG.SCALE_H_FLIP = False
    c.OUTPUT_DIR = "."
    c.DTYPE = "float32"  # :559
    # build-specific switches (not in the reference)
    c.ODW = CN()
    c.ODW.LOSS_IMPL = "fused"     # "fused": selection logic on the device; "loops": straight-line restatement
    # data-parallel gradient exchange (engine.GradExchange): element type on the wire -- "fp32" (the reference's DDP) or
    # "bf16" (half the xGMI bytes; gradients rounded once before the sum, fp32 again in the optimiser) -- and the
    # number of row blocks a large Linear's weight gradient is produced AND exchanged in
    c.ODW.GRAD_EXCHANGE = "fp32"
    c.ODW.WGRAD_SLICES = 4
    # HIP graphs of the body: how many input shapes stay captured at once (least recently used evicted; each holds its
    # body's activations, ~1 GB at VOC sizes); 0 = always eager
    c.ODW.GRAPH_CACHE = 4
    # device-resident loss lists (weak_head/loss_device.py): capacity, per image, of the IoU-sampled rows of a step -- the
    # stacked views' operand is allocated for twice that many rows (150 KB each at VGG16 / 7 x 7); a step that samples more
    # raises (loudly, one or two steps later)
    c.ODW.MAX_SAMPLED_ROWS = 4096
    c.SEED = -1
    c.min_size = 20                                           # :550
    return c
