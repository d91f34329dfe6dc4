"""yacs/detectron2 CfgNode stand-in: attribute access, yaml `_BASE_` chains, merge_from_list, freeze.
get_cfg() carries the detectron2 defaults of exactly the keys the Cube R-CNN path reads
(values copied from detectron2's documented defaults [third-party])."""
import ast
import copy
import os

import yaml

BASE_KEY = "_BASE_"


class CfgNode(dict):
    def __init__(self, init_dict=None, new_allowed=False):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        object.__setattr__(self, "_new_allowed", new_allowed)
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self._frozen:
            raise AttributeError(f"Attempted to set {name} to {value}, but CfgNode is immutable")
        self[name] = value

    def freeze(self):
        object.__setattr__(self, "_frozen", True)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def defrost(self):
        object.__setattr__(self, "_frozen", False)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    def is_frozen(self):
        return self._frozen

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        n = CfgNode(new_allowed=self._new_allowed)
        for k, v in self.items():
            dict.__setitem__(n, k, copy.deepcopy(v, memo))
        return n

    # ---- merging
    @staticmethod
    def _load_yaml_with_base(filename):
        with open(filename, "r") as f:
            cfg = yaml.safe_load(f) or {}
        if BASE_KEY in cfg:
            base = cfg.pop(BASE_KEY)
            if base.startswith("~"):
                base = os.path.expanduser(base)
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(filename), base)
            base_cfg = CfgNode._load_yaml_with_base(base)
            CfgNode._merge_dicts(cfg, base_cfg)
            return base_cfg
        return cfg

    @staticmethod
    def _merge_dicts(a, b):
        """merge a into b"""
        for k, v in a.items():
            if isinstance(v, dict) and k in b and isinstance(b[k], dict):
                CfgNode._merge_dicts(v, b[k])
            else:
                b[k] = v

    def merge_from_file(self, filename, allow_unsafe=False):
        loaded = CfgNode._load_yaml_with_base(filename)
        self._merge_from(loaded, [])

    def merge_from_other_cfg(self, other):
        self._merge_from(other, [])

    def _merge_from(self, d, path):
        for k, v in d.items():
            full = ".".join(path + [k])
            if isinstance(v, dict):
                if k not in self:
                    if self._new_allowed or k in ("PLOT", "TRAIN"):
                        self[k] = CfgNode(v, new_allowed=True)
                        continue
                    raise KeyError(f"Non-existent config key: {full}")
                if not isinstance(self[k], CfgNode):
                    raise KeyError(f"config key {full} is not a node")
                self[k]._merge_from(v, path + [k])
            else:
                if k not in self and not self._new_allowed:
                    raise KeyError(f"Non-existent config key: {full}")
                self[k] = self._coerce(v, self.get(k), full)

    @staticmethod
    def _coerce(new, old, key):
        if isinstance(new, str):
            try:
                lit = ast.literal_eval(new)
                if old is None or not isinstance(old, str):
                    new = lit
            except (ValueError, SyntaxError):
                pass
        if old is None or new is None:
            return new
        if isinstance(old, tuple) and isinstance(new, list):
            return tuple(new)
        if isinstance(old, list) and isinstance(new, tuple):
            return list(new)
        if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
            return float(new)
        if isinstance(old, bool) != isinstance(new, bool) and not isinstance(old, (str, list, tuple)):
            if isinstance(old, bool) or isinstance(new, bool):
                raise ValueError(f"Type mismatch for {key}: {type(old)} vs {type(new)}")
        return new

    def merge_from_list(self, cfg_list):
        assert len(cfg_list) % 2 == 0, "Override list has odd length"
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            d = self
            parts = full_key.split(".")
            for p in parts[:-1]:
                if p not in d:
                    raise KeyError(f"Non-existent key: {full_key}")
                d = d[p]
            if parts[-1] not in d and not d._new_allowed:
                raise KeyError(f"Non-existent key: {full_key}")
            d[parts[-1]] = self._coerce(v, d.get(parts[-1]), full_key)

    def dump(self):
        def conv(n):
            return {k: conv(v) if isinstance(v, CfgNode) else (list(v) if isinstance(v, tuple) else v)
                    for k, v in n.items()}
        return yaml.safe_dump(conv(self))


CN = CfgNode


def get_cfg():
    """detectron2 defaults (subset used by the Cube R-CNN path)."""
    _C = CN()
    _C.VERSION = 2
    _C.MODEL = CN()
    _C.MODEL.LOAD_PROPOSALS = False
    _C.MODEL.MASK_ON = False
    _C.MODEL.KEYPOINT_ON = False
    _C.MODEL.DEVICE = "cuda"
    _C.MODEL.META_ARCHITECTURE = "GeneralizedRCNN"
    _C.MODEL.WEIGHTS = ""
    _C.MODEL.PIXEL_MEAN = [103.530, 116.280, 123.675]
    _C.MODEL.PIXEL_STD = [1.0, 1.0, 1.0]
    _C.INPUT = CN()
    _C.INPUT.MIN_SIZE_TRAIN = (800,)
    _C.INPUT.MIN_SIZE_TRAIN_SAMPLING = "choice"
    _C.INPUT.MAX_SIZE_TRAIN = 1333
    _C.INPUT.MIN_SIZE_TEST = 800
    _C.INPUT.MAX_SIZE_TEST = 1333
    _C.INPUT.RANDOM_FLIP = "horizontal"
    _C.INPUT.FORMAT = "BGR"
    _C.DATASETS = CN()
    _C.DATASETS.TRAIN = ()
    _C.DATASETS.TEST = ()
    _C.DATALOADER = CN()
    _C.DATALOADER.NUM_WORKERS = 4
    _C.DATALOADER.ASPECT_RATIO_GROUPING = True
    _C.DATALOADER.SAMPLER_TRAIN = "TrainingSampler"
    _C.DATALOADER.REPEAT_THRESHOLD = 0.0
    _C.DATALOADER.FILTER_EMPTY_ANNOTATIONS = True
    _C.MODEL.BACKBONE = CN()
    _C.MODEL.BACKBONE.NAME = "build_resnet_backbone"
    _C.MODEL.BACKBONE.FREEZE_AT = 2
    _C.MODEL.FPN = CN()
    _C.MODEL.FPN.IN_FEATURES = []
    _C.MODEL.FPN.OUT_CHANNELS = 256
    _C.MODEL.FPN.NORM = ""
    _C.MODEL.FPN.FUSE_TYPE = "sum"
    _C.MODEL.PROPOSAL_GENERATOR = CN()
    _C.MODEL.PROPOSAL_GENERATOR.NAME = "RPN"
    _C.MODEL.PROPOSAL_GENERATOR.MIN_SIZE = 0
    _C.MODEL.ANCHOR_GENERATOR = CN()
    _C.MODEL.ANCHOR_GENERATOR.NAME = "DefaultAnchorGenerator"
    _C.MODEL.ANCHOR_GENERATOR.SIZES = [[32, 64, 128, 256, 512]]
    _C.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS = [[0.5, 1.0, 2.0]]
    _C.MODEL.ANCHOR_GENERATOR.ANGLES = [[-90, 0, 90]]
    _C.MODEL.ANCHOR_GENERATOR.OFFSET = 0.0
    _C.MODEL.RPN = CN()
    _C.MODEL.RPN.HEAD_NAME = "StandardRPNHead"
    _C.MODEL.RPN.IN_FEATURES = ["res4"]
    _C.MODEL.RPN.BOUNDARY_THRESH = -1
    _C.MODEL.RPN.IOU_THRESHOLDS = [0.3, 0.7]
    _C.MODEL.RPN.IOU_LABELS = [0, -1, 1]
    _C.MODEL.RPN.BATCH_SIZE_PER_IMAGE = 256
    _C.MODEL.RPN.POSITIVE_FRACTION = 0.5
    _C.MODEL.RPN.BBOX_REG_LOSS_TYPE = "smooth_l1"
    _C.MODEL.RPN.BBOX_REG_LOSS_WEIGHT = 1.0
    _C.MODEL.RPN.BBOX_REG_WEIGHTS = (1.0, 1.0, 1.0, 1.0)
    _C.MODEL.RPN.SMOOTH_L1_BETA = 0.0
    _C.MODEL.RPN.LOSS_WEIGHT = 1.0
    _C.MODEL.RPN.PRE_NMS_TOPK_TRAIN = 12000
    _C.MODEL.RPN.PRE_NMS_TOPK_TEST = 6000
    _C.MODEL.RPN.POST_NMS_TOPK_TRAIN = 2000
    _C.MODEL.RPN.POST_NMS_TOPK_TEST = 1000
    _C.MODEL.RPN.NMS_THRESH = 0.7
    _C.MODEL.RPN.CONV_DIMS = [-1]
    _C.MODEL.ROI_HEADS = CN()
    _C.MODEL.ROI_HEADS.NAME = "Res5ROIHeads"
    _C.MODEL.ROI_HEADS.NUM_CLASSES = 80
    _C.MODEL.ROI_HEADS.IN_FEATURES = ["res4"]
    _C.MODEL.ROI_HEADS.IOU_THRESHOLDS = [0.5]
    _C.MODEL.ROI_HEADS.IOU_LABELS = [0, 1]
    _C.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 512
    _C.MODEL.ROI_HEADS.POSITIVE_FRACTION = 0.25
    _C.MODEL.ROI_HEADS.SCORE_THRESH_TEST = 0.05
    _C.MODEL.ROI_HEADS.NMS_THRESH_TEST = 0.5
    _C.MODEL.ROI_HEADS.PROPOSAL_APPEND_GT = True
    _C.MODEL.ROI_BOX_HEAD = CN()
    _C.MODEL.ROI_BOX_HEAD.NAME = ""
    _C.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE = "smooth_l1"
    _C.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT = 1.0
    _C.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS = (10.0, 10.0, 5.0, 5.0)
    _C.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA = 0.0
    _C.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION = 14
    _C.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO = 0
    _C.MODEL.ROI_BOX_HEAD.POOLER_TYPE = "ROIAlignV2"
    _C.MODEL.ROI_BOX_HEAD.NUM_FC = 0
    _C.MODEL.ROI_BOX_HEAD.FC_DIM = 1024
    _C.MODEL.ROI_BOX_HEAD.NUM_CONV = 0
    _C.MODEL.ROI_BOX_HEAD.CONV_DIM = 256
    _C.MODEL.ROI_BOX_HEAD.NORM = ""
    _C.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = False
    _C.MODEL.ROI_BOX_HEAD.TRAIN_ON_PRED_BOXES = False
    _C.MODEL.RESNETS = CN()
    _C.MODEL.RESNETS.DEPTH = 50
    _C.MODEL.RESNETS.OUT_FEATURES = ["res4"]
    _C.MODEL.RESNETS.NORM = "FrozenBN"
    _C.SOLVER = CN()
    _C.SOLVER.LR_SCHEDULER_NAME = "WarmupMultiStepLR"
    _C.SOLVER.MAX_ITER = 40000
    _C.SOLVER.BASE_LR = 0.001
    _C.SOLVER.MOMENTUM = 0.9
    _C.SOLVER.NESTEROV = False
    _C.SOLVER.WEIGHT_DECAY = 0.0001
    _C.SOLVER.WEIGHT_DECAY_NORM = 0.0
    _C.SOLVER.GAMMA = 0.1
    _C.SOLVER.STEPS = (30000,)
    _C.SOLVER.WARMUP_FACTOR = 1.0 / 1000
    _C.SOLVER.WARMUP_ITERS = 1000
    _C.SOLVER.WARMUP_METHOD = "linear"
    _C.SOLVER.CHECKPOINT_PERIOD = 5000
    _C.SOLVER.IMS_PER_BATCH = 16
    _C.SOLVER.BIAS_LR_FACTOR = 1.0
    _C.SOLVER.WEIGHT_DECAY_BIAS = None
    _C.SOLVER.CLIP_GRADIENTS = CN({"ENABLED": False, "CLIP_TYPE": "value", "CLIP_VALUE": 1.0, "NORM_TYPE": 2.0})
    _C.TEST = CN()
    _C.TEST.EVAL_PERIOD = 0
    _C.TEST.DETECTIONS_PER_IMAGE = 100
    _C.OUTPUT_DIR = "./output"
    _C.SEED = -1
    _C.VIS_PERIOD = 0
    return _C
