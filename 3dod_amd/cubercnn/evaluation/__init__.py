from .omni3d_evaluation import Omni3DParams, Omni3Deval, iou_xywh
