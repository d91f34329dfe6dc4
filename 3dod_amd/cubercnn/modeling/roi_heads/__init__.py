from .cube_head import CubeHead, ROI_CUBE_HEAD_REGISTRY, build_cube_head
from .fast_rcnn import FastRCNNOutputs, fast_rcnn_inference
from .roi_heads import ROIHeads3D, build_roi_heads
from .roi_heads_score import ROIHeads3DScore
from .boxer import ROIHeads_Boxer
