"""CubeHead -- cubercnn/modeling/roi_heads/cube_head.py:24-202 (same parameter names / init).
Input is the NHWC-flattened ROIAlign output (n, 7*7*256) in the activation dtype; the first FC's weight keeps the
reference's (c,y,x) column order in the state dict and is permuted to (y,x,c) in its compute copy.
FC layers run through ops.linear / ops.linear_cat (f32: the implicit-GEMM kernels on the f32 MFMA)."""
from typing import Dict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ....d2lite import Registry, ShapeSpec
from .... import hipops as ops
from ...util.math_util import rotation_6d_to_matrix
from ..backbone.fpn import c2_xavier_fill

ROI_CUBE_HEAD_REGISTRY = Registry("ROI_CUBE_HEAD")
bf16 = torch.bfloat16


def fc_nhwc(x_nhwc_flat, fc, chw, relu=False):
    """Linear (+ ReLU in the GEMM epilogue) whose weight columns are in (c,h,w) order applied to an (h,w,c)-flattened input."""
    return ops.linear(x_nhwc_flat, fc.weight, fc.bias, chw=tuple(chw), relu=relu)


@ROI_CUBE_HEAD_REGISTRY.register()
class CubeHead(nn.Module):
    def __init__(self, cfg, input_shape: ShapeSpec):
        super().__init__()
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        self.use_conf = cfg.MODEL.ROI_CUBE_HEAD.USE_CONFIDENCE
        self.z_type = cfg.MODEL.ROI_CUBE_HEAD.Z_TYPE
        self.pose_type = cfg.MODEL.ROI_CUBE_HEAD.POSE_TYPE
        self.cluster_bins = cfg.MODEL.ROI_CUBE_HEAD.CLUSTER_BINS
        self.shared_fc = cfg.MODEL.ROI_CUBE_HEAD.SHARED_FC
        if cfg.MODEL.ROI_CUBE_HEAD.NUM_CONV > 0:
            raise ValueError("built: the cube head without convolutions (configs/Base.yaml); got NUM_CONV {}".format(
                cfg.MODEL.ROI_CUBE_HEAD.NUM_CONV))
        num_fc = cfg.MODEL.ROI_CUBE_HEAD.NUM_FC
        fc_dim = cfg.MODEL.ROI_CUBE_HEAD.FC_DIM
        self._in_chw = (input_shape.channels, input_shape.height, input_shape.width)
        self._output_size = self._in_chw
        # SHARED_FC = False (cube_head.py:56-111): one FC trunk per predictor, created in the reference's order (dims, XY, pose,
        # Z, confidence per layer) so that a seeded initialisation matches
        if self.shared_fc:
            self.feature_generator = nn.Sequential()
            trunks = [self.feature_generator]
        else:
            self.feature_generator_XY = nn.Sequential()
            self.feature_generator_dims = nn.Sequential()
            self.feature_generator_pose = nn.Sequential()
            self.feature_generator_Z = nn.Sequential()
            trunks = [self.feature_generator_dims, self.feature_generator_XY, self.feature_generator_pose, self.feature_generator_Z]
            if self.use_conf:
                self.feature_generator_conf = nn.Sequential()
                trunks.append(self.feature_generator_conf)
        for k in range(num_fc):
            fc_dim_in = int(np.prod(self._output_size))
            self._output_size = fc_dim
            for gen in trunks:
                fc = nn.Linear(fc_dim_in, fc_dim)
                c2_xavier_fill(fc)
                gen.add_module("fc{}".format(k + 1), fc)
                gen.add_module("fc_relu{}".format(k + 1), nn.ReLU())
        self.bbox_3D_dims = nn.Linear(self._output_size, self.num_classes * 3)
        nn.init.normal_(self.bbox_3D_dims.weight, std=0.001)
        nn.init.constant_(self.bbox_3D_dims.bias, 0)
        self.bbox_3D_center_deltas = nn.Linear(self._output_size, self.num_classes * 2)
        nn.init.normal_(self.bbox_3D_center_deltas.weight, std=0.001)
        nn.init.constant_(self.bbox_3D_center_deltas.bias, 0)
        if self.pose_type == '6d':
            self.bbox_3D_pose = nn.Linear(self._output_size, self.num_classes * 6)
        else:
            raise ValueError('Cuboid pose type {} is not recognized'.format(self.pose_type))
        nn.init.normal_(self.bbox_3D_pose.weight, std=0.001)
        nn.init.constant_(self.bbox_3D_pose.bias, 0)
        self.bbox_3D_center_depth = nn.Linear(self._output_size, self.num_classes * max(1, self.cluster_bins))    # [bin][class] (cube_head.py:141,196-197)
        nn.init.normal_(self.bbox_3D_center_depth.weight, std=0.001)
        nn.init.constant_(self.bbox_3D_center_depth.bias, 1)
        if self.use_conf:
            self.bbox_3D_uncertainty = nn.Linear(self._output_size, self.num_classes * 1)
            nn.init.normal_(self.bbox_3D_uncertainty.weight, std=0.001)
            nn.init.constant_(self.bbox_3D_uncertainty.bias, 5)

    def _trunk(self, gen, x):
        fcs = [m for m in gen if isinstance(m, nn.Linear)]
        h = fc_nhwc(x, fcs[0], self._in_chw, relu=True)
        for fc in fcs[1:]:
            h = ops.linear(h, fc.weight, fc.bias, relu=True)
        return h

    def _separate(self, x):
        """SHARED_FC = False (cube_head.py:170-178): every predictor on its own trunk -> [deltas, dims, pose6, z, (uncert)]"""
        pairs = [(self.feature_generator_XY, self.bbox_3D_center_deltas), (self.feature_generator_dims, self.bbox_3D_dims),
                 (self.feature_generator_pose, self.bbox_3D_pose), (self.feature_generator_Z, self.bbox_3D_center_depth)]
        if self.use_conf:
            pairs.append((self.feature_generator_conf, self.bbox_3D_uncertainty))
        outs = []
        for gen, m in pairs:                  # (linear_cat pads the predictor's rows to the GEMM tile)
            y, offs = ops.linear_cat(self._trunk(gen, x), [m.weight], [m.bias])
            outs.append(y[:, offs[0]:offs[1]].float())
        return outs

    def forward(self, x):
        """x (n, H*W*C) bf16 in (h,w,c) order -> (deltas (n,K,2), z (n,K,1), dims (n,K,3), pose (n,K,3,3), uncert (n,K))."""
        n = x.shape[0]
        if self.shared_fc:
            h = self._trunk(self.feature_generator, x)
            preds = [self.bbox_3D_center_deltas, self.bbox_3D_dims, self.bbox_3D_pose, self.bbox_3D_center_depth]
            if self.use_conf:
                preds.append(self.bbox_3D_uncertainty)
            y, offs = ops.linear_cat(h, [m.weight for m in preds], [m.bias for m in preds])      # the predictors as one GEMM
            outs = [y[:, offs[i]:offs[i + 1]].float().contiguous() for i in range(len(preds))]
        else:
            outs = [t.contiguous() for t in self._separate(x)]
        box_2d_deltas, box_dims, box_pose, box_z = outs[:4]
        box_uncert = outs[4].clip(0.01) if self.use_conf else None
        box_pose = rotation_6d_to_matrix(box_pose.view(-1, 6))
        box_2d_deltas = box_2d_deltas.view(n, self.num_classes, 2)
        box_dims = box_dims.view(n, self.num_classes, 3)
        box_pose = box_pose.view(n, self.num_classes, 3, 3)
        box_z = box_z.view(n, self.cluster_bins, self.num_classes, -1) if self.cluster_bins > 1 else box_z.view(n, self.num_classes, -1)
        return box_2d_deltas, box_z, box_dims, box_pose, box_uncert


def _forward_fused(self, x):
    """training form for the static-shape path: the five predictors as ONE GEMM.  Returns (raw (n, 13K) f32, layout) with
    layout = column offsets of [deltas 2K, dims 3K, pose6d 6K, z K, uncert K]; the per-class gather, the 6D -> matrix
    conversion and the uncertainty clip happen in ops.cube_head_loss (only for each RoI's own class)."""
    assert self.use_conf
    K = self.num_classes
    if not self.shared_fc:
        # per-predictor trunks: five GEMM chains, their outputs laid side by side in the fused layout
        return torch.cat(self._separate(x), 1), (0, 2 * K, 5 * K, 11 * K, (11 + max(1, self.cluster_bins)) * K)
    h = self._trunk(self.feature_generator, x)
    preds = [self.bbox_3D_center_deltas, self.bbox_3D_dims, self.bbox_3D_pose, self.bbox_3D_center_depth,
             self.bbox_3D_uncertainty]
    y, offs = ops.linear_cat(h, [m.weight for m in preds], [m.bias for m in preds])
    assert tuple(offs[:5]) == (0, 2 * K, 5 * K, 11 * K, (11 + max(1, self.cluster_bins)) * K)
    return y, tuple(offs[:5])                            # y (n, 13K rounded up to 16) f32; consumers take its row stride


CubeHead.forward_fused = _forward_fused


def build_cube_head(cfg, input_shape: ShapeSpec):
    name = cfg.MODEL.ROI_CUBE_HEAD.NAME
    return ROI_CUBE_HEAD_REGISTRY.get(name)(cfg, input_shape)
