"""Extra config keys of the reference -- cubercnn/config/config.py:4-187 (same names and defaults)."""
from ...d2lite.config import CfgNode as CN


def get_cfg_defaults(cfg):
    cfg.DATASETS.CATEGORY_NAMES = []
    cfg.DATASETS.IGNORE_NAMES = []
    cfg.DATALOADER.BALANCE_DATASETS = False
    cfg.DATASETS.TRUNCATION_THRES = 0.99
    cfg.DATASETS.VISIBILITY_THRES = 0.01
    cfg.DATASETS.MIN_HEIGHT_THRES = 0.00
    cfg.DATASETS.MAX_DEPTH = 1e8
    cfg.DATASETS.MODAL_2D_BOXES = False
    cfg.DATASETS.TRUNC_2D_BOXES = True
    cfg.MODEL.RPN.IGNORE_THRESHOLD = 0.5

    cfg.MODEL.ROI_CUBE_HEAD = CN()
    cfg.MODEL.ROI_CUBE_HEAD.NAME = "CubeHead"
    cfg.MODEL.ROI_CUBE_HEAD.POOLER_RESOLUTION = 7
    cfg.MODEL.ROI_CUBE_HEAD.POOLER_SAMPLING_RATIO = 0
    cfg.MODEL.ROI_CUBE_HEAD.POOLER_TYPE = "ROIAlignV2"
    cfg.MODEL.ROI_CUBE_HEAD.NUM_CONV = 0
    cfg.MODEL.ROI_CUBE_HEAD.CONV_DIM = 256
    cfg.MODEL.ROI_CUBE_HEAD.NUM_FC = 2
    cfg.MODEL.ROI_CUBE_HEAD.FC_DIM = 1024
    cfg.MODEL.ROI_CUBE_HEAD.NUMBER_OF_PROPOSALS = 1000
    cfg.MODEL.ROI_CUBE_HEAD.Z_TYPE = "direct"
    cfg.MODEL.ROI_CUBE_HEAD.POSE_TYPE = "6d"
    cfg.MODEL.ROI_CUBE_HEAD.INVERSE_Z_WEIGHT = False
    cfg.MODEL.ROI_CUBE_HEAD.VIRTUAL_DEPTH = True
    cfg.MODEL.ROI_CUBE_HEAD.VIRTUAL_FOCAL = 512.0
    cfg.MODEL.ROI_CUBE_HEAD.DISENTANGLED_LOSS = True
    cfg.MODEL.ROI_CUBE_HEAD.CLUSTER_BINS = 1
    cfg.MODEL.USE_BN = True
    cfg.MODEL.ROI_CUBE_HEAD.ALLOCENTRIC_POSE = True
    cfg.MODEL.ROI_CUBE_HEAD.CHAMFER_POSE = True
    cfg.MODEL.ROI_CUBE_HEAD.SHARED_FC = True
    cfg.MODEL.STABILIZE = 0.01
    cfg.MODEL.ROI_CUBE_HEAD.DIMS_PRIORS_ENABLED = True
    cfg.MODEL.ROI_CUBE_HEAD.DIMS_PRIORS_FUNC = "exp"
    cfg.MODEL.ROI_CUBE_HEAD.USE_CONFIDENCE = 1.0
    cfg.MODEL.ROI_CUBE_HEAD.LOSS_W_3D = 1.0
    cfg.MODEL.ROI_CUBE_HEAD.LOSS_W_XY = 1.0
    cfg.MODEL.ROI_CUBE_HEAD.LOSS_W_POSE = 7.0
    cfg.MODEL.ROI_CUBE_HEAD.LOSS_W_NORMAL_VEC = 20.0
    cfg.MODEL.ROI_CUBE_HEAD.LOSS_W_IOU = 1.0
    cfg.MODEL.ROI_CUBE_HEAD.LOSS_W_SEG = 2.5
    cfg.MODEL.ROI_CUBE_HEAD.LOSS_W_Z = 1.0
    cfg.MODEL.ROI_CUBE_HEAD.LOSS_W_DIMS = 20.0
    cfg.MODEL.ROI_CUBE_HEAD.LOSS_W_DEPTH = 1.0
    cfg.MODEL.DLA = CN()
    cfg.MODEL.DLA.TYPE = "dla34"
    cfg.MODEL.DLA.TRICKS = False
    cfg.MODEL.ROI_CUBE_HEAD.LOSS_W_JOINT = 1.0
    cfg.SOLVER.TYPE = "sgd"
    cfg.MODEL.RESNETS.TORCHVISION = True
    cfg.TEST.DETECTIONS_PER_IMAGE = 100
    cfg.TEST.VISIBILITY_THRES = 1 / 2.0
    cfg.TEST.TRUNCATION_THRES = 1 / 2.0
    cfg.INPUT.RANDOM_FLIP = "horizontal"
    cfg.MODEL.RPN.OBJECTNESS_UNCERTAINTY = "IoUness"
    cfg.MODEL.ROI_CUBE_HEAD.SCALE_ROI_BOXES = 0.0
    cfg.MODEL.WEIGHTS_PRETRAIN = ""
    cfg.MODEL.ROI_CUBE_HEAD.TEST = "bas"
    cfg.MODEL.ROI_CUBE_HEAD.DIMS_PRIORS_PRECOMPUTED = False
    cfg.PLOT = CN(new_allowed=True)
    cfg.PLOT.OUTPUT_DIR = ""
    cfg.PLOT.EVAL = ""
    cfg.PLOT.MODE2D = ""
    cfg.PLOT.SCORING_FUNC = None
    cfg.PLOT.PROPOSAL_FUNC = None
    cfg.PLOT.number_of_proposals = 1000
    cfg.TRAIN = CN(new_allowed=True)
    cfg.TRAIN.pseudo_gt = "learn"
    cfg.log = True
    cfg.loss_functions = ["dims", "pose_alignment", "pose_ground", "iou", "z", "z_pseudo_gt_patch", "depth"]
    cfg.MODEL.DEPTH_ON = False
