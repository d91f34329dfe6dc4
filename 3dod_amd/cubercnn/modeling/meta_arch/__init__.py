from .rcnn3d import RCNN3D, build_model, build_backbone
