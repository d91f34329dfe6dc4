"""DLA-34 trunk on the HIP conv kernels.  Same module tree (hence the same state-dict keys,
e.g. `level2.tree1.conv1.weight`) as cubercnn/modeling/backbone/dla.py:40-68,156-321,417-507 of the
reference; nn.Conv2d / nn.BatchNorm2d are used as PARAMETER CONTAINERS only -- the arithmetic runs in
cr_conv2d_* / cr_bn_* (NHWC bf16 activations, f32 statistics).  BatchNorm is per-GPU (dla.py:17)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ....d2lite import BACKBONE_REGISTRY, ShapeSpec
from .... import hipops as ops
from .fpn import FPN, Backbone, to_channels_last

BatchNorm = nn.BatchNorm2d


def _conv_bn(x, conv, bn, relu, residual=None):
    w = conv.weight
    if w.shape[1] < x.shape[-1]:      # RGB stem: activations carry 8 (bf16) or 4 (f32) channels, 3 real + zeros
        w = ops.pad_input_channels(w, x.shape[-1])
    return ops.conv_bn_act(x, w, bn.weight, bn.bias, bn.running_mean, bn.running_var, stride=conv.stride[0],
                           pad=conv.padding[0], relu=relu, residual=residual, eps=bn.eps, momentum=bn.momentum,
                           training=bn.training)


class BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1, dilation=1):
        super().__init__()
        assert dilation == 1
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=dilation, bias=False)
        self.bn1 = BatchNorm(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=dilation, bias=False)
        self.bn2 = BatchNorm(planes)
        self.stride = stride

    def forward(self, x, residual=None):
        if residual is None:
            residual = x
        out = _conv_bn(x, self.conv1, self.bn1, relu=True)
        return _conv_bn(out, self.conv2, self.bn2, relu=True, residual=residual)     # out += residual; relu


class Root(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, residual):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, 1, stride=1, bias=False, padding=(kernel_size - 1) // 2)
        self.bn = BatchNorm(out_channels)
        self.relu = nn.ReLU(inplace=True)
        self.residual = residual

    def forward(self, *x):
        children = x
        if not self.residual and self.bn.training and self.conv.kernel_size[0] == 1:
            # training: the children's gradients come from per-child backward-data GEMMs (ops._RootConvBN), not from the
            # gradient of the concatenation
            return ops.root_conv_bn_act(children, self.conv.weight, self.bn.weight, self.bn.bias, self.bn.running_mean,
                                        self.bn.running_var, relu=True, eps=self.bn.eps, momentum=self.bn.momentum, training=True)
        cat = torch.cat(x, 3)                                        # channel concat (NHWC)
        return _conv_bn(cat, self.conv, self.bn, relu=True, residual=children[0] if self.residual else None)


class _Project(nn.Sequential):
    def forward(self, x):
        return _conv_bn(x, self[0], self[1], relu=False)


class _MaxPool2(nn.Module):
    def forward(self, x):
        return ops.maxpool2x2(x)


class Tree(nn.Module):
    def __init__(self, levels, block, in_channels, out_channels, stride=1, level_root=False, root_dim=0,
                 root_kernel_size=1, dilation=1, root_residual=False):
        super().__init__()
        if root_dim == 0:
            root_dim = 2 * out_channels
        if level_root:
            root_dim += in_channels
        if levels == 1:
            self.tree1 = block(in_channels, out_channels, stride, dilation=dilation)
            self.tree2 = block(out_channels, out_channels, 1, dilation=dilation)
        else:
            self.tree1 = Tree(levels - 1, block, in_channels, out_channels, stride, root_dim=0,
                              root_kernel_size=root_kernel_size, dilation=dilation, root_residual=root_residual)
            self.tree2 = Tree(levels - 1, block, out_channels, out_channels, root_dim=root_dim + out_channels,
                              root_kernel_size=root_kernel_size, dilation=dilation, root_residual=root_residual)
        if levels == 1:
            self.root = Root(root_dim, out_channels, root_kernel_size, root_residual)
        self.level_root = level_root
        self.root_dim = root_dim
        self.downsample = None
        self.project = None
        self.levels = levels
        if stride > 1:
            assert stride == 2
            self.downsample = _MaxPool2()
        if in_channels != out_channels:
            self.project = _Project(nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, bias=False),
                                    BatchNorm(out_channels))

    def forward(self, x, residual=None, children=None):
        children = [] if children is None else children
        bottom = self.downsample(x) if self.downsample else x
        residual = self.project(bottom) if self.project else bottom
        if self.level_root:
            children.append(bottom)
        x1 = self.tree1(x, residual)
        if self.levels == 1:
            x2 = self.tree2(x1)
            x = self.root(x2, x1, *children)
        else:
            children.append(x1)
            x = self.tree2(x1, children=children)
        return x


class _ConvLevel(nn.Sequential):
    """(conv, bn, relu) triples, dla.py:287-297."""

    def forward(self, x):
        mods = list(self)
        for i in range(0, len(mods), 3):
            x = _conv_bn(x, mods[i], mods[i + 1], relu=True)
        return x


class DLA(nn.Module):
    def __init__(self, levels, channels, num_classes=1000, block=BasicBlock, residual_root=False, return_levels=False,
                 pool_size=7, linear_root=False):
        super().__init__()
        self.channels = channels
        self.base_layer = _ConvLevel(nn.Conv2d(3, channels[0], kernel_size=7, stride=1, padding=3, bias=False),
                                     BatchNorm(channels[0]), nn.ReLU(inplace=True))
        self.level0 = self._make_conv_level(channels[0], channels[0], levels[0])
        self.level1 = self._make_conv_level(channels[0], channels[1], levels[1], stride=2)
        self.level2 = Tree(levels[2], block, channels[1], channels[2], 2, level_root=False, root_residual=residual_root)
        self.level3 = Tree(levels[3], block, channels[2], channels[3], 2, level_root=True, root_residual=residual_root)
        self.level4 = Tree(levels[4], block, channels[3], channels[4], 2, level_root=True, root_residual=residual_root)
        self.level5 = Tree(levels[5], block, channels[4], channels[5], 2, level_root=True, root_residual=residual_root)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, BatchNorm):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_conv_level(self, inplanes, planes, convs, stride=1, dilation=1):
        modules = []
        for i in range(convs):
            modules.extend([nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride if i == 0 else 1, padding=dilation,
                                      bias=False, dilation=dilation), BatchNorm(planes), nn.ReLU(inplace=True)])
            inplanes = planes
        return _ConvLevel(*modules)


def dla34(pretrained=False, tricks=False, **kwargs):
    if pretrained:
        # dla.py:300-309 downloads ImageNet weights; there is no network here -> random init
        pass
    return DLA([1, 1, 1, 2, 2, 1], [16, 32, 64, 128, 256, 512], block=BasicBlock, **kwargs)


class DLABackbone(Backbone):
    def __init__(self, cfg, input_shape, pretrained=True):
        super().__init__()
        if cfg.MODEL.DLA.TYPE != "dla34":
            raise ValueError("only dla34 is built (the BASELINE configuration); got {}".format(cfg.MODEL.DLA.TYPE))
        base = dla34(pretrained=False, tricks=cfg.MODEL.DLA.TRICKS)
        self._out_feature_channels = {'p2': 64, 'p3': 128, 'p4': 256, 'p5': 512, 'p6': 512}
        self.base_layer = base.base_layer
        self.level0 = base.level0
        self.level1 = base.level1
        self.level2 = base.level2
        self.level3 = base.level3
        self.level4 = base.level4
        self.level5 = base.level5
        self._out_feature_strides = {'p2': 4, 'p3': 8, 'p4': 16, 'p5': 32, 'p6': 64}
        self._out_features = ['p2', 'p3', 'p4', 'p5', 'p6']
        to_channels_last(self)

    def forward(self, x):
        outputs = {}
        base_layer = self.base_layer(x)
        level0 = self.level0(base_layer)
        level1 = self.level1(level0)
        level2 = self.level2(level1)
        level3 = self.level3(level2)
        level4 = self.level4(level3)
        level5 = self.level5(level4)
        level6 = ops.subsample2x(level5)            # F.max_pool2d(level5, kernel_size=1, stride=2), dla.py:474
        outputs['p2'] = level2
        outputs['p3'] = level3
        outputs['p4'] = level4
        outputs['p5'] = level5
        outputs['p6'] = level6
        return outputs


@BACKBONE_REGISTRY.register()
def build_dla_from_vision_fpn_backbone(cfg, input_shape: ShapeSpec, priors=None):
    """dla.py:484-507."""
    bottom_up = DLABackbone(cfg, input_shape, pretrained=False)
    return FPN(bottom_up=bottom_up, in_features=cfg.MODEL.FPN.IN_FEATURES, out_channels=cfg.MODEL.FPN.OUT_CHANNELS,
               norm=cfg.MODEL.FPN.NORM, fuse_type=cfg.MODEL.FPN.FUSE_TYPE)
