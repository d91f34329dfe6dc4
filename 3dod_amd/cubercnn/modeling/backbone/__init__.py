from .dla import DLA, DLABackbone, dla34, build_dla_from_vision_fpn_backbone
from .fpn import FPN, Backbone, to_channels_last
from .fpn import LastLevelMaxPool
from .resnet import ResNet, build_resnet_from_vision_fpn_backbone
