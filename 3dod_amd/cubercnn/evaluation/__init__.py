from .omni3d_evaluation import Omni3DParams, Omni3Deval, iou_xywh, instances_to_coco_json, Omni3DEvaluator, \
    Omni3DEvaluationHelper, inference_on_dataset
