from .backbone import *  # noqa: F401,F403  (registers build_dla_from_vision_fpn_backbone)
from .proposal_generator import RPNWithIgnore  # noqa: F401
from .roi_heads import ROIHeads3D, ROIHeads3DScore, ROIHeads_Boxer, CubeHead, build_roi_heads  # noqa: F401
from .meta_arch import RCNN3D, RCNN3D_combined_features, BoxNet, build_model, build_backbone  # noqa: F401
