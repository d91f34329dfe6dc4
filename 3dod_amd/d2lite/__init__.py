"""Minimal stand-ins for the detectron2 pieces the Cube R-CNN hot path is written against
(detectron2 is not installed here nor on the GPU box): config node with yaml `_BASE_`
inheritance, registries, Boxes / Instances / ImageList / ShapeSpec, box coding, matcher,
anchor generator, event storage.  Each restates detectron2's documented behaviour
[third-party, absent from the reference tree -- SURVEY.md 8c "parity unpinned"].
If the real detectron2 is importable, INTEGRATION.md shows how to register into it instead.
"""
from .config import CfgNode, get_cfg
from .registry import Registry, META_ARCH_REGISTRY, BACKBONE_REGISTRY, PROPOSAL_GENERATOR_REGISTRY, \
    ROI_HEADS_REGISTRY, ROI_BOX_HEAD_REGISTRY, RPN_HEAD_REGISTRY, ANCHOR_GENERATOR_REGISTRY
from .structures import Boxes, Instances, ImageList, ShapeSpec, pairwise_iou, pairwise_ioa, cat
from .box_ops import Box2BoxTransform, Matcher, DefaultAnchorGenerator, subsample_labels_d2
from .events import EventStorage, get_event_storage, JSONWriter, CommonMetricPrinter
