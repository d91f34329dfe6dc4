"""detectron2 Backbone / FPN stand-ins on the HIP conv kernels [third-party behaviour restated;
wired by the reference at cubercnn/modeling/backbone/dla.py:484-507].  Features are NHWC bf16."""
import math

import torch
import torch.nn as nn

from ....d2lite import ShapeSpec
from .... import hipops as ops


def c2_xavier_fill(module):
    """fvcore c2_xavier_fill: kaiming_uniform_(a=1), zero bias."""
    nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def to_channels_last(module):
    """conv weights live in channels_last storage = the [Cout][k*k][Cin] layout the kernels read; the
    logical (Cout,Cin,k,k) shape (and so the state dict) is unchanged."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return module


class Backbone(nn.Module):
    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
                for name in self._out_features}

    @property
    def size_divisibility(self):
        return 0

    @property
    def padding_constraints(self):
        return {}


class LastLevelMaxPool(nn.Module):
    """detectron2 LastLevelMaxPool [third-party]: one extra level = max_pool2d(kernel 1, stride 2) of "p5"."""

    def __init__(self):
        super().__init__()
        self.num_levels = 1
        self.in_feature = "p5"

    def forward(self, x):
        return [ops.subsample2x(x)]


class FPN(Backbone):
    """lateral 1x1 + top-down nearest-2x sum + output 3x3, no norm; optional top block (LastLevelMaxPool)."""

    def __init__(self, bottom_up, in_features, out_channels, norm="", top_block=None, fuse_type="sum"):
        super().__init__()
        assert norm == "", "only the configuration used by the reference is built"
        assert fuse_type in ("sum", "avg")
        input_shapes = bottom_up.output_shape()
        strides = [input_shapes[f].stride for f in in_features]
        in_channels_per_feature = [input_shapes[f].channels for f in in_features]
        lateral_convs, output_convs = [], []
        for idx, in_channels in enumerate(in_channels_per_feature):
            lateral_conv = nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=True)
            output_conv = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True)
            c2_xavier_fill(lateral_conv)
            c2_xavier_fill(output_conv)
            stage = int(math.log2(strides[idx]))
            self.add_module("fpn_lateral{}".format(stage), lateral_conv)
            self.add_module("fpn_output{}".format(stage), output_conv)
            lateral_convs.append(lateral_conv)
            output_convs.append(output_conv)
        self.lateral_convs = lateral_convs[::-1]
        self.output_convs = output_convs[::-1]
        self.in_features = tuple(in_features)
        self.bottom_up = bottom_up
        self.top_block = top_block
        self._out_feature_strides = {"p{}".format(int(math.log2(s))): s for s in strides}
        if top_block is not None:                    # detectron2 FPN: extra levels named after the last stage
            stage = int(math.log2(strides[-1]))
            for s_ in range(stage, stage + top_block.num_levels):
                self._out_feature_strides["p{}".format(s_ + 1)] = 2 ** (s_ + 1)
        self._out_features = list(self._out_feature_strides.keys())
        self._out_feature_channels = {k: out_channels for k in self._out_features}
        self._size_divisibility = strides[-1]
        self._fuse_type = fuse_type
        to_channels_last(self)

    @property
    def size_divisibility(self):
        return self._size_divisibility

    @property
    def padding_constraints(self):
        return {"square_size": 0}

    def forward(self, x):
        """x: NHWC bf16 image batch (channels padded to 8) -> dict name -> NHWC bf16 feature map."""
        bottom_up_features = self.bottom_up(x)
        # top-down pathway first (lateral 1x1 + nearest-2x sum, coarse to fine), then ALL output convolutions in one grouped
        # launch per direction (ops.conv_bias_act_group): they are independent of each other, and alone the coarse levels
        # are a handful of tiles each
        lat0 = self.lateral_convs[0]
        prev = ops.conv_bias_act(bottom_up_features[self.in_features[-1]], lat0.weight, lat0.bias, 1, 0)
        inner = [prev]
        for idx, lateral_conv in enumerate(self.lateral_convs):
            if idx > 0:
                features = bottom_up_features[self.in_features[-idx - 1]]
                lateral = ops.conv_bias_act(features, lateral_conv.weight, lateral_conv.bias, 1, 0)
                prev = ops.upsample2x_add(lateral, prev)
                if self._fuse_type == "avg":
                    prev = prev / 2
                inner.append(prev)
        outs = ops.conv_bias_act_group(inner, [c.weight for c in self.output_convs], [c.bias for c in self.output_convs], pad=1)
        results = outs[::-1]                                   # fine -> coarse
        if self.top_block is not None:
            src = bottom_up_features[self.top_block.in_feature] if self.top_block.in_feature in bottom_up_features \
                else results[self._out_features.index(self.top_block.in_feature)]
            results.extend(self.top_block(src))
        assert len(self._out_features) == len(results)
        return {f: res for f, res in zip(self._out_features, results)}
