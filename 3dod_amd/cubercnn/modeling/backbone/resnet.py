"""ResNet-18/34 trunk (torchvision BasicBlock layout) + FPN with LastLevelMaxPool on the HIP conv kernels --
cubercnn/modeling/backbone/resnet.py:12-96 of the reference, which takes the modules from torchvision.models.resnet34
[third-party, absent here: structure, initialisation and state-dict keys (`layer2.0.downsample.0.weight`, ...) restated
from its public definition -> parity unpinned w.r.t. torchvision].  nn.Conv2d / nn.BatchNorm2d are parameter containers;
the arithmetic runs in cr_conv2d_* / cr_bn_* / cr_maxpool3x3s2_* (NHWC bf16)."""
import torch.nn as nn

from ....d2lite import BACKBONE_REGISTRY, ShapeSpec
from .... import hipops as ops
from .fpn import FPN, Backbone, LastLevelMaxPool, to_channels_last
from .dla import _conv_bn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x if self.downsample is None else _conv_bn(x, self.downsample[0], self.downsample[1], relu=False)
        out = _conv_bn(x, self.conv1, self.bn1, relu=True)
        return _conv_bn(out, self.conv2, self.bn2, relu=True, residual=identity)       # out += identity; relu


class _Layer(nn.Sequential):
    def forward(self, x):
        for blk in self:
            x = blk(x)
        return x


_LAYERS = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3]}


class ResNet(Backbone):
    """resnet.py:12-63: conv1/bn1/relu/maxpool, layer1..4 -> p2..p5, p6 = max_pool2d(p5, 1, stride 2)."""

    def __init__(self, cfg, input_shape, pretrained=False):
        super().__init__()
        depth = cfg.MODEL.RESNETS.DEPTH
        if depth not in _LAYERS:
            raise ValueError('No configuration currently supporting depth of {}'.format(depth))
        self._out_feature_channels = {'p2': 64, 'p3': 128, 'p4': 256, 'p5': 512, 'p6': 512}
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.inplanes = 64
        n = _LAYERS[depth]
        self.layer1 = self._make_layer(64, n[0], 1)
        self.layer2 = self._make_layer(128, n[1], 2)
        self.layer3 = self._make_layer(256, n[2], 2)
        self.layer4 = self._make_layer(512, n[3], 2)
        for m in self.modules():                      # torchvision's initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._out_feature_strides = {'p2': 4, 'p3': 8, 'p4': 16, 'p5': 32, 'p6': 64}
        self._out_features = ['p2', 'p3', 'p4', 'p5', 'p6']
        to_channels_last(self)

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return _Layer(*layers)

    def forward(self, x):
        x = _conv_bn(x, self.conv1, self.bn1, relu=True)
        x = ops.maxpool3x3s2(x)
        p2 = self.layer1(x)
        p3 = self.layer2(p2)
        p4 = self.layer3(p3)
        p5 = self.layer4(p4)
        p6 = ops.subsample2x(p5)                       # F.max_pool2d(p5, kernel_size=1, stride=2), resnet.py:55
        return {'p2': p2, 'p3': p3, 'p4': p4, 'p5': p5, 'p6': p6}


@BACKBONE_REGISTRY.register()
def build_resnet_from_vision_fpn_backbone(cfg, input_shape: ShapeSpec, priors=None):
    """resnet.py:66-96 (TORCHVISION route; ImageNet weights cannot be fetched offline -> random init)."""
    if not cfg.MODEL.RESNETS.TORCHVISION:
        raise ValueError("only the torchvision ResNet route of the reference's configs is built")
    bottom_up = ResNet(cfg, input_shape, pretrained=False)
    return FPN(bottom_up=bottom_up, in_features=cfg.MODEL.FPN.IN_FEATURES, out_channels=cfg.MODEL.FPN.OUT_CHANNELS,
               norm=cfg.MODEL.FPN.NORM, top_block=LastLevelMaxPool(), fuse_type=cfg.MODEL.FPN.FUSE_TYPE)
