from .rcnn3d import RCNN3D, RCNN3D_combined_features, BoxNet, build_model, build_backbone
