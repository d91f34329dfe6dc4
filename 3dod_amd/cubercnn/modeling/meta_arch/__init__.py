from .rcnn3d import RCNN3D, BoxNet, build_model, build_backbone
