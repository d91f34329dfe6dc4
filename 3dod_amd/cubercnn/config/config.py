"""Extra config keys of Cube R-CNN on top of the detectron2 defaults: same key names and default values as
cubercnn/config/config.py:4-187 of the reference (the yaml files and command-line overrides address them by name),
kept as one nested table that is merged into the config tree."""
from ...d2lite.config import CfgNode as CN

_OPEN = object()          # marks a node that accepts new keys from yaml (CfgNode(new_allowed=True))

EXTRA_DEFAULTS = {
    "DATASETS": {
        "CATEGORY_NAMES": [], "IGNORE_NAMES": [],
        # filters applied when the Omni3D annotations are loaded
        "TRUNCATION_THRES": 0.99, "VISIBILITY_THRES": 0.01, "MIN_HEIGHT_THRES": 0.00, "MAX_DEPTH": 1e8,
        "MODAL_2D_BOXES": False, "TRUNC_2D_BOXES": True,
    },
    "DATALOADER": {"BALANCE_DATASETS": False},
    "INPUT": {"RANDOM_FLIP": "horizontal"},
    "SOLVER": {"TYPE": "sgd"},
    "TEST": {"DETECTIONS_PER_IMAGE": 100, "VISIBILITY_THRES": 0.5, "TRUNCATION_THRES": 0.5},
    "MODEL": {
        "USE_BN": True, "STABILIZE": 0.01, "WEIGHTS_PRETRAIN": "", "DEPTH_ON": False,
        "RPN": {"IGNORE_THRESHOLD": 0.5, "OBJECTNESS_UNCERTAINTY": "IoUness"},
        "DLA": {"TYPE": "dla34", "TRICKS": False},
        "RESNETS": {"TORCHVISION": True},
        "ROI_CUBE_HEAD": {
            "NAME": "CubeHead", "TEST": "bas",
            # RoI pooling and the shared FC trunk
            "POOLER_RESOLUTION": 7, "POOLER_SAMPLING_RATIO": 0, "POOLER_TYPE": "ROIAlignV2", "SCALE_ROI_BOXES": 0.0,
            "NUM_CONV": 0, "CONV_DIM": 256, "NUM_FC": 2, "FC_DIM": 1024, "SHARED_FC": True, "CLUSTER_BINS": 1,
            # parametrisation of the 3D box
            "Z_TYPE": "direct", "POSE_TYPE": "6d", "VIRTUAL_DEPTH": True, "VIRTUAL_FOCAL": 512.0,
            "ALLOCENTRIC_POSE": True, "DIMS_PRIORS_ENABLED": True, "DIMS_PRIORS_FUNC": "exp",
            "DIMS_PRIORS_PRECOMPUTED": False, "NUMBER_OF_PROPOSALS": 1000,
            # losses
            "DISENTANGLED_LOSS": True, "CHAMFER_POSE": True, "INVERSE_Z_WEIGHT": False, "USE_CONFIDENCE": 1.0,
            "LOSS_W_3D": 1.0, "LOSS_W_XY": 1.0, "LOSS_W_Z": 1.0, "LOSS_W_DIMS": 20.0, "LOSS_W_POSE": 7.0,
            "LOSS_W_JOINT": 1.0,
            # weak (2D-only supervision) losses
            "LOSS_W_NORMAL_VEC": 20.0, "LOSS_W_IOU": 1.0, "LOSS_W_SEG": 2.5, "LOSS_W_DEPTH": 1.0,
        },
    },
    "PLOT": {_OPEN: True, "OUTPUT_DIR": "", "EVAL": "", "MODE2D": "", "SCORING_FUNC": None, "PROPOSAL_FUNC": None,
             "number_of_proposals": 1000},
    "TRAIN": {_OPEN: True, "pseudo_gt": "learn"},
    "log": True,
    "loss_functions": ["dims", "pose_alignment", "pose_ground", "iou", "z", "z_pseudo_gt_patch", "depth"],
}


def _merge(node, table):
    for key, val in table.items():
        if key is _OPEN:
            continue
        if isinstance(val, dict):
            if key not in node:
                node[key] = CN(new_allowed=True) if val.get(_OPEN) else CN()
            _merge(node[key], val)
        else:
            node[key] = val


def get_cfg_defaults(cfg):
    _merge(cfg, EXTRA_DEFAULTS)
    return cfg
