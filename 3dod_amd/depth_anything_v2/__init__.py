"""Depth-Anything-V2 (metric) forward on MI355X: DINOv2 ViT encoder + DPT head with the module names and state-dict keys
of the reference's `depth/metric_depth/depth_anything_v2` package, inference only."""
from .dinov2 import DINOv2, DinoVisionTransformer  # noqa: F401
from .dpt import DepthAnythingV2, DPTHead  # noqa: F401
