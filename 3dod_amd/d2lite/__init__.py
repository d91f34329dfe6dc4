"""Minimal stand-ins for the detectron2 pieces the Cube R-CNN hot path is written against
(detectron2 is not installed here nor on the GPU box): config node with yaml `_BASE_`
inheritance, registries, Boxes / Instances / ImageList / ShapeSpec, box coding, matcher,
anchor generator, event storage.  Each restates detectron2's documented behaviour
[third-party, absent from the reference tree -- SURVEY.md 8c "parity unpinned"].
If the real detectron2 is importable, INTEGRATION.md shows how to register into it instead.
"""


def _cr_bootstrap():
    """This file is executing as the TOP-LEVEL package `d2lite` (PYTHONPATH=<repo>/3dod_amd, the reference's layout):
    load the enclosing directory as the package `3dod_amd` and become an alias of `3dod_amd.d2lite`."""
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = sys.modules.get("3dod_amd")
    if pkg is None:
        spec = importlib.util.spec_from_file_location("3dod_amd", os.path.join(root, "__init__.py"),
                                                      submodule_search_locations=[root])
        pkg = importlib.util.module_from_spec(spec)
        sys.modules["3dod_amd"] = pkg
        spec.loader.exec_module(pkg)
    pkg._adopt_toplevel("d2lite")


if __name__ == "d2lite":
    _cr_bootstrap()
else:
    from .config import CfgNode, get_cfg
    from .registry import Registry, META_ARCH_REGISTRY, BACKBONE_REGISTRY, PROPOSAL_GENERATOR_REGISTRY, \
        ROI_HEADS_REGISTRY, ROI_BOX_HEAD_REGISTRY, RPN_HEAD_REGISTRY, ANCHOR_GENERATOR_REGISTRY
    from .structures import Boxes, Instances, ImageList, ShapeSpec, pairwise_iou, pairwise_ioa, cat
    from .box_ops import Box2BoxTransform, Matcher, DefaultAnchorGenerator, subsample_labels_d2
    from .events import EventStorage, get_event_storage, JSONWriter, CommonMetricPrinter
