"""DPT fusion blocks, forward only, NHWC bf16 (reference: depth/metric_depth/depth_anything_v2/util/blocks.py).
Every convolution is the library's MFMA implicit-GEMM kernel with bias / ReLU / residual fused in its epilogue."""
import torch
import torch.nn as nn

from ... import hipops as ops


def conv(x, m, relu=False, residual=None, out_f32=False):
    """nn.Conv2d module `m` (3x3 or 1x1, stride 1 / 2) on NHWC bf16 `x`"""
    k, s, p = m.kernel_size[0], m.stride[0], m.padding[0]
    w = m.weight
    if not w.is_contiguous(memory_format=torch.channels_last) and k > 1:
        raise RuntimeError("conv weights must be channels_last (DepthAnythingV2 converts them at construction / load)")
    wb, _ = ops.prepared_weights(w, False)
    b = m.bias.detach().float().contiguous() if m.bias is not None else None
    return ops.conv_fwd_raw(x, wb, w.shape[0], k, s, p, bias=b, residual=residual, relu=relu, out_f32=out_f32)


def _make_scratch(in_shape, out_shape, groups=1, expand=False):
    """blocks.py:4-29"""
    assert groups == 1 and not expand
    scratch = nn.Module()
    scratch.layer1_rn = nn.Conv2d(in_shape[0], out_shape, kernel_size=3, stride=1, padding=1, bias=False)
    scratch.layer2_rn = nn.Conv2d(in_shape[1], out_shape, kernel_size=3, stride=1, padding=1, bias=False)
    scratch.layer3_rn = nn.Conv2d(in_shape[2], out_shape, kernel_size=3, stride=1, padding=1, bias=False)
    if len(in_shape) >= 4:
        scratch.layer4_rn = nn.Conv2d(in_shape[3], out_shape, kernel_size=3, stride=1, padding=1, bias=False)
    return scratch


class ResidualConvUnit(nn.Module):
    """blocks.py:32-86: x + conv2(relu(conv1(relu(x))))"""

    def __init__(self, features, activation=None, bn=False):
        super().__init__()
        if bn:
            raise NotImplementedError("use_bn is False in every Depth-Anything-V2 configuration")
        self.bn = bn
        self.groups = 1
        self.conv1 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
        self.conv2 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)

    def forward(self, x):
        return conv(conv(torch.relu(x), self.conv1, relu=True), self.conv2, residual=x)


class FeatureFusionBlock(nn.Module):
    """blocks.py:89-148"""

    def __init__(self, features, activation=None, deconv=False, bn=False, expand=False, align_corners=True, size=None):
        super().__init__()
        assert not deconv and not expand and align_corners
        self.deconv, self.align_corners, self.groups, self.expand, self.size = deconv, align_corners, 1, expand, size
        self.out_conv = nn.Conv2d(features, features, kernel_size=1, stride=1, padding=0, bias=True)
        self.resConfUnit1 = ResidualConvUnit(features, activation, bn)
        self.resConfUnit2 = ResidualConvUnit(features, activation, bn)

    def forward(self, *xs, size=None):
        output = xs[0]
        if len(xs) == 2:
            output = output + self.resConfUnit1(xs[1])
        output = self.resConfUnit2(output)
        if size is None and self.size is None:
            size = (output.shape[1] * 2, output.shape[2] * 2)
        elif size is None:
            size = self.size
        output = ops.resize_bilinear_ac(output, size)
        return conv(output, self.out_conv)
