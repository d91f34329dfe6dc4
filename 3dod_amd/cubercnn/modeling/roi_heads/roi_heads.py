"""ROIHeads3D -- the stock Cube R-CNN 2D+3D head, cubercnn/modeling/roi_heads/roi_heads.py:1948-2851 of the
reference, on top of a detectron2 StandardROIHeads stand-in [third-party, restated].

Heavy ops: ROIAlign over the FPN pyramid (cr_roi_align_*), FC layers (library GEMM, bf16).  The per-RoI
decode and the disentangled corner losses are small float32 torch expressions on the device (n <= 128 FG
RoIs per image)."""
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ....d2lite import (ROI_HEADS_REGISTRY, ROI_BOX_HEAD_REGISTRY, Boxes, Instances, ShapeSpec, Matcher, cat,
                        pairwise_iou, pairwise_ioa, get_event_storage)
from .... import hipops as ops
from ...util import math_util as util
from ..backbone.fpn import c2_xavier_fill
from .cube_head import build_cube_head, fc_nhwc
from .fast_rcnn import FastRCNNOutputs

E_CONSTANT = 2.71828183
SQRT_2_CONSTANT = 1.41421356
bf16 = torch.bfloat16


def build_roi_heads(cfg, input_shape, priors=None):
    """roi_heads.py:72-77."""
    name = cfg.MODEL.ROI_HEADS.NAME
    return ROI_HEADS_REGISTRY.get(name)(cfg, input_shape, priors=priors)


class ROIPooler(nn.Module):
    """detectron2 ROIPooler(ROIAlignV2) [third-party]: level assignment + aligned ROIAlign, fused in one kernel.
    Output is (R, out, out, C) bf16 -- NHWC, flattened by the heads in (h,w,c) order."""

    def __init__(self, output_size, scales, sampling_ratio, pooler_type):
        super().__init__()
        assert pooler_type == "ROIAlignV2" and sampling_ratio == 0, "only the reference's pooler config is built"
        self.output_size = output_size
        self.scales = tuple(scales)

    def forward(self, x: List[torch.Tensor], box_lists: List[Boxes]):
        rois = cat([torch.cat([b.tensor.new_full((len(b), 1), i), b.tensor], 1) for i, b in enumerate(box_lists)], 0)
        return ops.roi_align_pyramid(x, rois, self.scales, self.output_size)


@ROI_BOX_HEAD_REGISTRY.register()
class FastRCNNConvFCHead(nn.Sequential):
    """detectron2 FastRCNNConvFCHead with NUM_CONV 0 [third-party]: fc1 -> relu -> fc2 -> relu."""

    def __init__(self, input_shape: ShapeSpec, *, conv_dims, fc_dims, conv_norm=""):
        super().__init__()
        assert len(conv_dims) == 0 and len(fc_dims) > 0
        self._in_chw = (input_shape.channels, input_shape.height, input_shape.width)
        self._output_size = self._in_chw
        self.fcs = []
        for k, fc_dim in enumerate(fc_dims):
            fc = nn.Linear(int(np.prod(self._output_size)), fc_dim)
            self.add_module("fc{}".format(k + 1), fc)
            self.add_module("fc_relu{}".format(k + 1), nn.ReLU())
            self.fcs.append(fc)
            self._output_size = fc_dim
        for layer in self.fcs:
            c2_xavier_fill(layer)

    def forward(self, x):
        """x (R, H, W, C) bf16."""
        h = fc_nhwc(x.flatten(1), self.fcs[0], self._in_chw, relu=True)
        for fc in self.fcs[1:]:
            h = ops.linear(h, fc.weight, fc.bias, relu=True)
        return h

    @property
    def output_shape(self):
        return ShapeSpec(channels=self._output_size)


def select_foreground_proposals(proposals, bg_label):
    fg_proposals, fg_selection_masks = [], []
    for proposals_per_image in proposals:
        gt_classes = proposals_per_image.gt_classes
        fg_selection_mask = (gt_classes != -1) & (gt_classes != bg_label)
        fg_idxs = fg_selection_mask.nonzero().squeeze(1)
        fg_proposals.append(proposals_per_image[fg_idxs])
        fg_selection_masks.append(fg_selection_mask)
    return fg_proposals, fg_selection_masks


class StandardROIHeads(nn.Module):
    def __init__(self, *, num_classes, batch_size_per_image, positive_fraction, proposal_matcher,
                 proposal_append_gt=True, box_in_features, box_pooler, box_head, box_predictor,
                 train_on_pred_boxes=False, **kwargs):
        super().__init__()
        self.num_classes = num_classes
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.proposal_matcher = proposal_matcher
        self.proposal_append_gt = proposal_append_gt
        self.in_features = self.box_in_features = box_in_features
        self.box_pooler = box_pooler
        self.box_head = box_head
        self.box_predictor = box_predictor
        self.train_on_pred_boxes = train_on_pred_boxes

    @classmethod
    def from_config(cls, cfg, input_shape):
        in_features = cfg.MODEL.ROI_HEADS.IN_FEATURES
        pooler_resolution = cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        pooler_scales = tuple(1.0 / input_shape[k].stride for k in in_features)
        in_channels = [input_shape[f].channels for f in in_features]
        assert len(set(in_channels)) == 1, in_channels
        in_channels = in_channels[0]
        box_pooler = ROIPooler(output_size=pooler_resolution, scales=pooler_scales,
                               sampling_ratio=cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO,
                               pooler_type=cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE)
        shape = ShapeSpec(channels=in_channels, height=pooler_resolution, width=pooler_resolution)
        head_cls = ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)
        box_head = head_cls(shape, conv_dims=[cfg.MODEL.ROI_BOX_HEAD.CONV_DIM] * cfg.MODEL.ROI_BOX_HEAD.NUM_CONV,
                            fc_dims=[cfg.MODEL.ROI_BOX_HEAD.FC_DIM] * cfg.MODEL.ROI_BOX_HEAD.NUM_FC,
                            conv_norm=cfg.MODEL.ROI_BOX_HEAD.NORM)
        return {
            "num_classes": cfg.MODEL.ROI_HEADS.NUM_CLASSES,
            "batch_size_per_image": cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE,
            "positive_fraction": cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION,
            "proposal_matcher": Matcher(cfg.MODEL.ROI_HEADS.IOU_THRESHOLDS, cfg.MODEL.ROI_HEADS.IOU_LABELS,
                                        allow_low_quality_matches=False),
            "proposal_append_gt": cfg.MODEL.ROI_HEADS.PROPOSAL_APPEND_GT,
            "box_in_features": in_features,
            "box_pooler": box_pooler,
            "box_head": box_head,
            "box_predictor": None,
            "train_on_pred_boxes": cfg.MODEL.ROI_BOX_HEAD.TRAIN_ON_PRED_BOXES,
        }


@ROI_HEADS_REGISTRY.register()
class ROIHeads3D(StandardROIHeads):
    def __init__(self, cfg, input_shape: Dict[str, ShapeSpec], priors=None):
        ret = StandardROIHeads.from_config(cfg, input_shape)
        ret["box_predictor"] = FastRCNNOutputs(cfg, ret["box_head"].output_shape)
        super().__init__(**ret)
        c = cfg.MODEL.ROI_CUBE_HEAD
        self.scale_roi_boxes = c.SCALE_ROI_BOXES
        self.allocentric_pose = c.ALLOCENTRIC_POSE
        self.chamfer_pose = c.CHAMFER_POSE
        self.virtual_depth = c.VIRTUAL_DEPTH
        self.virtual_focal = c.VIRTUAL_FOCAL
        self.loss_w_3d = c.LOSS_W_3D
        self.loss_w_xy = c.LOSS_W_XY
        self.loss_w_z = c.LOSS_W_Z
        self.loss_w_dims = c.LOSS_W_DIMS
        self.loss_w_pose = c.LOSS_W_POSE
        self.loss_w_joint = c.LOSS_W_JOINT
        self.disentangled_loss = c.DISENTANGLED_LOSS
        self.inverse_z_weight = c.INVERSE_Z_WEIGHT
        self.test_scale = cfg.INPUT.MIN_SIZE_TEST
        self.ignore_thresh = cfg.MODEL.RPN.IGNORE_THRESHOLD
        self.z_type = c.Z_TYPE
        self.pose_type = c.POSE_TYPE
        self.use_confidence = c.USE_CONFIDENCE
        self.cluster_bins = c.CLUSTER_BINS
        self.dims_priors_enabled = c.DIMS_PRIORS_ENABLED
        self.dims_priors_func = c.DIMS_PRIORS_FUNC
        if self.z_type != "direct" or self.cluster_bins > 1 or not self.disentangled_loss:
            raise ValueError("only Z_TYPE 'direct', CLUSTER_BINS 1 and the disentangled loss of configs/Base.yaml are built")
        if self.loss_w_3d > 0:
            in_features = cfg.MODEL.ROI_HEADS.IN_FEATURES
            pooler_scales = tuple(1.0 / input_shape[k].stride for k in in_features)
            self.cube_pooler = ROIPooler(output_size=c.POOLER_RESOLUTION, scales=pooler_scales,
                                         sampling_ratio=c.POOLER_SAMPLING_RATIO, pooler_type=c.POOLER_TYPE)
            in_channels = [input_shape[f].channels for f in in_features][0]
            shape = ShapeSpec(channels=in_channels, width=c.POOLER_RESOLUTION, height=c.POOLER_RESOLUTION)
            self.cube_head = build_cube_head(cfg, shape)
            if self.dims_priors_enabled and priors is not None:
                self.priors_dims_per_cat = nn.Parameter(torch.FloatTensor(priors['priors_dims_per_cat']).unsqueeze(0))
            else:
                self.priors_dims_per_cat = nn.Parameter(torch.ones(1, self.num_classes, 2, 3))
            self.priors_z_scales = nn.Parameter(torch.ones(self.num_classes, self.cluster_bins))

    # ------------------------------------------------------------------ forward
    def forward(self, images, features, proposals, Ks, im_scales_ratio, targets=None):
        """roi_heads.py:2116-2157.  images: ImageList (only sizes are used)."""
        im_dims = [tuple(s) for s in images.image_sizes]
        if self.training:
            # Training runs on the static-shape path (modeling/dense_train.py behind RCNN3D.dense_train): matching, ignore
            # rule and IoU-weighted sampling are fused kernels there.  The per-image Instances-list formulation of
            # label_and_sample_proposals (roi_heads.py:2737-2840 of the reference) is test infrastructure: it lives in
            # oracle/list_path.py and is attached by oracle.list_path.install(roi_heads) (tests only).
            if not hasattr(self, "label_and_sample_proposals"):
                raise RuntimeError("ROIHeads3D.forward(training) on proposal lists: use the static-shape path "
                                   "(model.dense_train = True); the list formulation is oracle/list_path.py")
            proposals = self.label_and_sample_proposals(proposals, targets)
            losses = self._forward_box(features, proposals)
            if self.loss_w_3d > 0:
                instances_3d, losses_cube = self._forward_cube(features, proposals, Ks, im_dims, im_scales_ratio)
                losses.update(losses_cube)
            else:
                instances_3d = None
            return instances_3d, losses
        if isinstance(proposals, list) and not np.any([isinstance(p, Instances) for p in proposals]):
            pred_instances = []
            for proposal, im_dim in zip(proposals, im_dims):
                pred_instances_i = Instances(im_dim)
                pred_instances_i.pred_boxes = Boxes(proposal['gt_bbox2D'])
                pred_instances_i.pred_classes = proposal['gt_classes']
                pred_instances_i.scores = torch.ones_like(proposal['gt_classes']).float()
                pred_instances.append(pred_instances_i)
        else:
            pred_instances = self._forward_box(features, proposals)
        if self.loss_w_3d > 0:
            pred_instances = self._forward_cube(features, pred_instances, Ks, im_dims, im_scales_ratio)
        return pred_instances, {}

    def _forward_box(self, features, proposals):
        """roi_heads.py:2160-2204."""
        feats = [features[f] for f in self.box_in_features]
        box_features = self.box_pooler(feats, [x.proposal_boxes for x in proposals])
        box_features = self.box_head(box_features)
        predictions = self.box_predictor(box_features)
        del box_features
        if self.training:
            losses = self.box_predictor.losses(predictions, proposals)
            pred_boxes = self.box_predictor.predict_boxes_for_gt_classes(predictions, proposals)
            for proposals_per_image, pred_boxes_per_image in zip(proposals, pred_boxes):
                proposals_per_image.pred_boxes = Boxes(pred_boxes_per_image)
            return losses
        pred_instances, _ = self.box_predictor.inference(predictions, proposals)
        return pred_instances

    def l1_loss(self, vals, target):
        return torch.abs(vals - target)          # F.smooth_l1_loss(beta=0, reduction='none'), roi_heads.py:2206-2207

    def chamfer_loss(self, vals, target):
        """roi_heads.py:2209-2215."""
        B = vals.shape[0]
        xx = vals.view(B, 8, 1, 3)
        yy = target.view(B, 1, 8, 3)
        l1_dist = (xx - yy).abs().sum(-1)
        return l1_dist.min(1).values.mean(-1) + l1_dist.min(2).values.mean(-1)

    def scale_proposals(self, proposal_boxes):
        """roi_heads.py:2217-2235 (including its use of the width for the height)."""
        if self.scale_roi_boxes > 0:
            out = []
            for boxes in proposal_boxes:
                centers = boxes.get_centers()
                widths = boxes.tensor[:, 2] - boxes.tensor[:, 0]
                heights = boxes.tensor[:, 2] - boxes.tensor[:, 0]
                x1 = centers[:, 0] - 0.5 * widths * self.scale_roi_boxes
                x2 = centers[:, 0] + 0.5 * widths * self.scale_roi_boxes
                y1 = centers[:, 1] - 0.5 * heights * self.scale_roi_boxes
                y2 = centers[:, 1] + 0.5 * heights * self.scale_roi_boxes
                out.append(Boxes(torch.stack([x1, y1, x2, y2], dim=1)))
            return out
        return proposal_boxes

    # ------------------------------------------------------------------ cube branch
    def _forward_cube(self, features, instances, Ks, im_current_dims, im_scales_ratio):
        """roi_heads.py:2237-2735 for the configuration of configs/Base.yaml (disentangled + chamfer + allocentric +
        virtual depth + exp dims priors + confidence)."""
        feats = [features[f] for f in self.in_features]
        if self.training:
            losses = {}
            self.normalize_factor = max(sum([i.gt_classes.numel() for i in instances]), 1.0)
            proposals, _ = select_foreground_proposals(instances, self.num_classes)
            proposal_boxes = [x.proposal_boxes for x in proposals]
            pred_boxes = [x.pred_boxes for x in proposals]
            box_classes = torch.cat([p.gt_classes for p in proposals], dim=0) if len(proposals) else torch.empty(0)
            gt_boxes3D = torch.cat([p.gt_boxes3D for p in proposals], dim=0)
            gt_poses = torch.cat([p.gt_poses for p in proposals], dim=0)
            assert len(gt_poses) == len(gt_boxes3D) == len(box_classes)
        else:
            proposals = instances
            pred_boxes = [x.pred_boxes for x in instances]
            proposal_boxes = pred_boxes
            box_classes = torch.cat([x.pred_classes for x in instances])
        proposal_boxes_scaled = self.scale_proposals(proposal_boxes)
        n = sum(len(b) for b in proposal_boxes_scaled)
        if n == 0:
            return instances if not self.training else (instances, {})
        if (not self.training) and box_classes.is_cuda and hasattr(ops, "cube_decode_infer") and self.use_confidence > 0 \
                and (self.dims_priors_func == 'exp' or not self.dims_priors_enabled):
            return self._infer_cube_fused(feats, instances, proposal_boxes_scaled, proposal_boxes, box_classes, Ks,
                                          im_current_dims, im_scales_ratio)
        cube_features = self.cube_pooler(feats, proposal_boxes_scaled).flatten(1)
        device = cube_features.device
        num_boxes_per_image = [len(i) for i in proposals]

        Ks_dev = [torch.as_tensor(K, dtype=torch.float32) for K in Ks]
        Ks_scaled_per_box = torch.cat([(Ks_dev[i] / im_scales_ratio[i]).unsqueeze(0).repeat([num, 1, 1])
                                       for (i, num) in enumerate(num_boxes_per_image)]).to(device)
        Ks_scaled_per_box[:, -1, -1] = 1
        focal_lengths_per_box = torch.cat([(Ks_dev[i][1, 1]).unsqueeze(0).repeat([num])
                                           for (i, num) in enumerate(num_boxes_per_image)]).to(device)
        im_ratios_per_box = torch.cat([torch.FloatTensor([im_scales_ratio[i]]).repeat(num)
                                       for (i, num) in enumerate(num_boxes_per_image)]).to(device)
        im_scales_per_box = torch.cat([torch.FloatTensor([im_current_dims[i][0]]).repeat(num)
                                       for (i, num) in enumerate(num_boxes_per_image)]).to(device)
        im_scales_original_per_box = im_scales_per_box * im_ratios_per_box
        if self.virtual_depth:
            virtual_to_real = util.compute_virtual_scale_from_focal_spaces(
                focal_lengths_per_box, im_scales_original_per_box, self.virtual_focal, im_scales_per_box)
        else:
            virtual_to_real = 1.0

        src_boxes = torch.cat([b.tensor for b in proposal_boxes], dim=0)
        src_widths = src_boxes[:, 2] - src_boxes[:, 0]
        src_heights = src_boxes[:, 3] - src_boxes[:, 1]
        src_ctr_x = src_boxes[:, 0] + 0.5 * src_widths
        src_ctr_y = src_boxes[:, 1] + 0.5 * src_heights

        cube_2d_deltas, cube_z, cube_dims, cube_pose, cube_uncert = self.cube_head(cube_features)
        fg_inds = torch.arange(n, device=device)
        cube_z = cube_z[fg_inds, box_classes, :]
        cube_dims = cube_dims[fg_inds, box_classes, :]
        cube_pose = cube_pose[fg_inds, box_classes, :, :]
        if self.use_confidence:
            cube_uncert = cube_uncert[fg_inds, box_classes]
        cube_2d_deltas = cube_2d_deltas[fg_inds, box_classes, :]

        fused = None
        if self.training and cube_2d_deltas.is_cuda and self.use_confidence > 0 and self.dims_priors_func == 'exp' \
                and hasattr(ops, "cube_decode_loss"):
            # K15/K16 in one kernel each way (cr_cube_loss_fwd / _bwd); the torch expressions below are the same
            # arithmetic and remain the path for eval-mode decode
            K = Ks_scaled_per_box
            K4 = torch.stack((K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]), 1)
            v2r = virtual_to_real if self.virtual_depth else torch.ones(n, device=device)
            pm = self.priors_dims_per_cat.detach()[0][box_classes][:, 0, :] if self.dims_priors_enabled \
                else torch.ones(n, 3, device=device)
            fused = ops.cube_decode_loss(cube_2d_deltas, cube_z[:, 0], cube_dims, cube_pose, cube_uncert, src_boxes, K4,
                                         v2r, pm, gt_boxes3D[:, :2], gt_boxes3D[:, 2], gt_boxes3D[:, 3:6], gt_poses,
                                         allocentric=self.allocentric_pose, chamfer_pose=self.chamfer_pose,
                                         use_conf=True, joint=self.loss_w_joint > 0)
        if fused is not None:
            return self._finish_fused_cube(fused, cube_uncert, gt_boxes3D, num_boxes_per_image, im_ratios_per_box,
                                           im_current_dims, box_classes, pred_boxes, n)
        cube_x = src_ctr_x + src_widths * cube_2d_deltas[:, 0]
        cube_y = src_ctr_y + src_heights * cube_2d_deltas[:, 1]
        cube_xy = torch.cat((cube_x.unsqueeze(1), cube_y.unsqueeze(1)), dim=1)
        cube_dims_norm = cube_dims
        if self.dims_priors_enabled:
            prior_dims = self.priors_dims_per_cat.detach()[0][box_classes]          # (n,2,3)
            prior_dims_mean = prior_dims[:, 0, :]
            prior_dims_std = prior_dims[:, 1, :]
            if self.dims_priors_func == 'sigmoid':
                prior_dims_min = (prior_dims_mean - 3 * prior_dims_std).clip(0.0)
                prior_dims_max = (prior_dims_mean + 3 * prior_dims_std)
                cube_dims = util.scaled_sigmoid(cube_dims_norm, min=prior_dims_min, max=prior_dims_max)
            elif self.dims_priors_func == 'exp':
                cube_dims = torch.exp(cube_dims_norm.clip(max=5)) * prior_dims_mean
        else:
            cube_dims = torch.exp(cube_dims_norm.clip(max=5))
        if self.allocentric_pose:
            cube_pose = util.R_from_allocentric(Ks_scaled_per_box, cube_pose, u=cube_x.detach(), v=cube_y.detach())
        cube_z = cube_z.squeeze(1)          # (n,) also for n == 1
        if self.virtual_depth:
            cube_z = cube_z * virtual_to_real

        if self.training:
            prefix = 'Cube/'
            storage = get_event_storage()
            K = Ks_scaled_per_box
            gt_2d = gt_boxes3D[:, :2]
            gt_z = gt_boxes3D[:, 2]
            gt_dims = gt_boxes3D[:, 3:6]
            gt_x3d = gt_z * (gt_2d[:, 0] - K[:, 0, 2]) / K[:, 0, 0]
            gt_y3d = gt_z * (gt_2d[:, 1] - K[:, 1, 2]) / K[:, 1, 1]
            gt_3d = torch.stack((gt_x3d, gt_y3d, gt_z)).T
            gt_box3d = torch.cat((gt_3d, gt_dims), dim=1)
            gt_corners = util.get_cuboid_verts_faces(gt_box3d, gt_poses)[0]

            # disentangled corner sets (roi_heads.py:2471-2508)
            cube_dis_x3d_from_z = cube_z * (gt_2d[:, 0] - K[:, 0, 2]) / K[:, 0, 0]
            cube_dis_y3d_from_z = cube_z * (gt_2d[:, 1] - K[:, 1, 2]) / K[:, 1, 1]
            cube_dis_z = torch.cat((torch.stack((cube_dis_x3d_from_z, cube_dis_y3d_from_z, cube_z)).T, gt_dims), dim=1)
            dis_z_corners = util.get_cuboid_verts_faces(cube_dis_z, gt_poses)[0]
            cube_dis_x3d = gt_z * (cube_x - K[:, 0, 2]) / K[:, 0, 0]
            cube_dis_y3d = gt_z * (cube_y - K[:, 1, 2]) / K[:, 1, 1]
            cube_dis_XY = torch.cat((torch.stack((cube_dis_x3d, cube_dis_y3d, gt_z)).T, gt_dims), dim=1)
            dis_XY_corners = util.get_cuboid_verts_faces(cube_dis_XY, gt_poses)[0]
            loss_xy = self.l1_loss(dis_XY_corners, gt_corners).contiguous().view(n, -1).mean(dim=1)
            dis_pose_corners = util.get_cuboid_verts_faces(gt_box3d, cube_pose)[0]
            dis_dims_corners = util.get_cuboid_verts_faces(torch.cat((gt_3d, cube_dims), dim=1), gt_poses)[0]
            loss_dims = self.l1_loss(dis_dims_corners, gt_corners).contiguous().view(n, -1).mean(dim=1)
            loss_z = self.l1_loss(dis_z_corners, gt_corners).contiguous().view(n, -1).mean(dim=1)
            if self.chamfer_pose:
                loss_pose = self.chamfer_loss(dis_pose_corners, gt_corners)
            else:
                loss_pose = self.l1_loss(dis_pose_corners, gt_corners).contiguous().view(n, -1).mean(dim=1)

            total_3D_loss_for_reporting = loss_dims * self.loss_w_dims
            total_3D_loss_for_reporting = total_3D_loss_for_reporting + loss_pose * self.loss_w_pose
            total_3D_loss_for_reporting = total_3D_loss_for_reporting + loss_xy * self.loss_w_xy
            total_3D_loss_for_reporting = total_3D_loss_for_reporting + loss_z * self.loss_w_z
            total_3D_loss_for_reporting = total_3D_loss_for_reporting.detach()

            if self.loss_w_joint > 0:
                cube_j_x3d = cube_z * (cube_x - K[:, 0, 2]) / K[:, 0, 0]
                cube_j_y3d = cube_z * (cube_y - K[:, 1, 2]) / K[:, 1, 1]
                cube_j = torch.cat((torch.stack((cube_j_x3d, cube_j_y3d, cube_z)).T, cube_dims), dim=1)
                dis_z_corners_joint = util.get_cuboid_verts_faces(cube_j, cube_pose)[0]
                if self.chamfer_pose and self.disentangled_loss:
                    loss_joint = self.chamfer_loss(dis_z_corners_joint, gt_corners)
                else:
                    loss_joint = self.l1_loss(dis_z_corners_joint, gt_corners).contiguous().view(n, -1).mean(dim=1)
                valid_joint = loss_joint < np.inf
                total_3D_loss_for_reporting = total_3D_loss_for_reporting + (loss_joint * self.loss_w_joint).detach()

            # tracking scalars stay on the device (no .item() host syncs, unlike roi_heads.py:2601-2606)
            with torch.no_grad():
                z_error = (cube_z - gt_z).abs()
                storage.put_scalar(prefix + 'z_error', z_error.mean(), smoothing_hint=False)
                storage.put_scalar(prefix + 'dims_error', (cube_dims - gt_dims).abs().mean(), smoothing_hint=False)
                storage.put_scalar(prefix + 'xy_error', (cube_xy - gt_2d).abs().mean(), smoothing_hint=False)
                storage.put_scalar(prefix + 'z_close', (z_error < 0.20).float().mean(), smoothing_hint=False)
                storage.put_scalar(prefix + 'total_3D_loss',
                                   self.loss_w_3d * self.safely_reduce_losses(total_3D_loss_for_reporting),
                                   smoothing_hint=False)

            if self.inverse_z_weight:
                inverse_z_w = 1 / torch.log(gt_boxes3D[:, 2].clip(E_CONSTANT))
                loss_dims = loss_dims * inverse_z_w
                loss_xy = loss_xy * inverse_z_w
                loss_z = loss_z * inverse_z_w
                loss_pose = loss_pose * inverse_z_w
                if self.loss_w_joint > 0:
                    loss_joint = loss_joint * inverse_z_w

            if self.use_confidence > 0:
                uncert_sf = SQRT_2_CONSTANT * torch.exp(-cube_uncert)
                loss_dims = loss_dims * uncert_sf
                loss_xy = loss_xy * uncert_sf
                loss_z = loss_z * uncert_sf
                loss_pose = loss_pose * uncert_sf
                if self.loss_w_joint > 0:
                    loss_joint = loss_joint * uncert_sf
                losses.update({prefix + 'uncert': self.use_confidence * self.safely_reduce_losses(cube_uncert.clone())})
                storage.put_scalar(prefix + 'conf', torch.exp(-cube_uncert.detach()).mean(), smoothing_hint=False)

            if self.loss_w_dims > 0:
                losses.update({prefix + 'loss_dims': self.safely_reduce_losses(loss_dims) * self.loss_w_dims * self.loss_w_3d})
            losses.update({prefix + 'loss_xy': self.safely_reduce_losses(loss_xy) * self.loss_w_xy * self.loss_w_3d})
            losses.update({prefix + 'loss_z': self.safely_reduce_losses(loss_z) * self.loss_w_z * self.loss_w_3d})
            losses.update({prefix + 'loss_pose': self.safely_reduce_losses(loss_pose) * self.loss_w_pose * self.loss_w_3d})
            if self.loss_w_joint > 0:
                # loss_joint[valid_joint] with `if valid_joint.any()` (roi_heads.py:2676-2677) without a host sync:
                # safely_reduce_losses already averages the finite entries only
                losses.update({prefix + 'loss_joint': self.safely_reduce_losses(
                    torch.where(valid_joint, loss_joint, torch.full_like(loss_joint, float('inf'))),
                    absent_if_none=True) * self.loss_w_joint * self.loss_w_3d})

        # ---- inference packing (roi_heads.py:2682-2735)
        if len(cube_z.shape) == 0:
            cube_z = cube_z.unsqueeze(0)
        K = Ks_scaled_per_box
        cube_x3d = cube_z * (cube_x - K[:, 0, 2]) / K[:, 0, 0]
        cube_y3d = cube_z * (cube_y - K[:, 1, 2]) / K[:, 1, 1]
        cube_3D = torch.cat((torch.stack((cube_x3d, cube_y3d, cube_z)).T, cube_dims,
                             cube_xy * im_ratios_per_box.unsqueeze(1)), dim=1)
        if self.use_confidence:
            cube_conf = torch.exp(-cube_uncert)
            cube_3D = torch.cat((cube_3D, cube_conf.unsqueeze(1)), dim=1)
        cube_3D = cube_3D.split(num_boxes_per_image)
        cube_pose = cube_pose.split(num_boxes_per_image)
        box_classes = box_classes.split(num_boxes_per_image)
        pred_instances = instances if not self.training else [Instances(image_size) for image_size in im_current_dims]
        for cube_3D_i, cube_pose_i, instances_i, box_classes_i, pred_boxes_i in \
                zip(cube_3D, cube_pose, pred_instances, box_classes, pred_boxes):
            if instances_i.has('scores'):
                instances_i.scores = (instances_i.scores * cube_3D_i[:, -1]) ** (1 / 2)
            else:
                instances_i.scores = cube_3D_i[:, -1]
            if not instances_i.has('pred_classes'):
                instances_i.pred_classes = box_classes_i
            if not instances_i.has('pred_boxes'):
                instances_i.pred_boxes = pred_boxes_i
            instances_i.pred_bbox3D = util.get_cuboid_verts_faces(cube_3D_i[:, :6], cube_pose_i)[0]
            instances_i.pred_center_cam = cube_3D_i[:, :3]
            instances_i.pred_center_2D = cube_3D_i[:, 6:8]
            instances_i.pred_dimensions = cube_3D_i[:, 3:6]
            instances_i.pred_pose = cube_pose_i
        if self.training:
            return pred_instances, losses
        return pred_instances

    def _infer_cube_fused(self, feats, instances, boxes_scaled, boxes, box_classes, Ks, im_current_dims, im_scales_ratio):
        """inference decode + packing (roi_heads.py:2353-2436, 2682-2735) on the fused path: one ROIAlign, the shared
        FCs + one predictor GEMM, one decode kernel (cr_cube_decode_infer); the torch expressions of _forward_cube stay
        the CPU / oracle statement of the same arithmetic (tests/test_gpu_model.py compares the two)."""
        dev = feats[0].device
        counts = [len(b) for b in boxes]
        n = sum(counts)
        idx = torch.repeat_interleave(torch.arange(len(counts), device=dev), torch.tensor(counts, device=dev))
        rois = torch.cat([idx[:, None].float(), torch.cat([b.tensor for b in boxes_scaled])], 1)
        cube_features = ops.roi_align_pyramid(feats, rois, self.cube_pooler.scales, self.cube_pooler.output_size).flatten(1)
        raw, layout = self.cube_head.forward_fused(cube_features)
        rows = []
        for k, r, d in zip(Ks, im_scales_ratio, im_current_dims):
            k = torch.as_tensor(k, dtype=torch.float32)
            v2r = util.compute_virtual_scale_from_focal_spaces(float(k[1, 1]), float(d[0]) * float(r), self.virtual_focal,
                                                               float(d[0])) if self.virtual_depth else 1.0
            rows.append([float(k[0, 0]) / r, float(k[1, 1]) / r, float(k[0, 2]) / r, float(k[1, 2]) / r, float(v2r), float(r)])
        meta6 = torch.tensor(rows, dtype=torch.float32).pin_memory().to(dev, non_blocking=True)
        priors = self.priors_dims_per_cat.detach()[0, :, 0, :].contiguous() if self.dims_priors_enabled else None
        out = ops.cube_decode_infer(raw, layout, self.num_classes, box_classes, idx, torch.cat([b.tensor for b in boxes]),
                                    meta6, priors, allocentric=self.allocentric_pose)
        for inst, o, cls_i in zip(instances, out.split(counts), box_classes.split(counts)):
            m = o.shape[0]
            inst.scores = (inst.scores * o[:, 8]) ** (1 / 2) if inst.has('scores') else o[:, 8]
            if not inst.has('pred_classes'):
                inst.pred_classes = cls_i
            inst.pred_bbox3D = o[:, 18:42].reshape(m, 8, 3)
            inst.pred_center_cam = o[:, 0:3]
            inst.pred_center_2D = o[:, 6:8]
            inst.pred_dimensions = o[:, 3:6]
            inst.pred_pose = o[:, 9:18].reshape(m, 3, 3)
        return instances

    def _finish_fused_cube(self, fused, cube_uncert, gt_boxes3D, num_boxes_per_image, im_ratios_per_box,
                           im_current_dims, box_classes, pred_boxes, n):
        """reductions, tracking scalars and the training-time Instances packing on top of the fused kernel's
        per-RoI outputs (roi_heads.py:2562-2735)."""
        L, dec = fused
        prefix = 'Cube/'
        storage = get_event_storage()
        losses = {}
        loss_dims, loss_xy, loss_z, loss_pose, loss_joint = L.unbind(1)
        cube_x3d_y3d_z = torch.stack((dec[:, 15], dec[:, 16], dec[:, 2]), 1)
        cube_z, cube_dims, cube_xy = dec[:, 2], dec[:, 3:6], dec[:, 0:2]
        cube_pose = dec[:, 6:15].reshape(n, 3, 3)
        gt_z, gt_dims, gt_2d = gt_boxes3D[:, 2], gt_boxes3D[:, 3:6], gt_boxes3D[:, :2]
        with torch.no_grad():
            sf = SQRT_2_CONSTANT * torch.exp(-cube_uncert)
            Lr = L / sf[:, None]
            total = Lr[:, 0] * self.loss_w_dims + Lr[:, 3] * self.loss_w_pose + Lr[:, 1] * self.loss_w_xy + \
                Lr[:, 2] * self.loss_w_z
            if self.loss_w_joint > 0:
                total = total + Lr[:, 4] * self.loss_w_joint
            z_error = (cube_z - gt_z).abs()
            storage.put_scalar(prefix + 'z_error', z_error.mean(), smoothing_hint=False)
            storage.put_scalar(prefix + 'dims_error', (cube_dims - gt_dims).abs().mean(), smoothing_hint=False)
            storage.put_scalar(prefix + 'xy_error', (cube_xy - gt_2d).abs().mean(), smoothing_hint=False)
            storage.put_scalar(prefix + 'z_close', (z_error < 0.20).float().mean(), smoothing_hint=False)
            storage.put_scalar(prefix + 'total_3D_loss', self.loss_w_3d * self.safely_reduce_losses(total),
                               smoothing_hint=False)
            storage.put_scalar(prefix + 'conf', torch.exp(-cube_uncert).mean(), smoothing_hint=False)
        if self.inverse_z_weight:
            inverse_z_w = 1 / torch.log(gt_z.clip(E_CONSTANT))
            loss_dims, loss_xy, loss_z, loss_pose, loss_joint = [t * inverse_z_w for t in
                                                                 (loss_dims, loss_xy, loss_z, loss_pose, loss_joint)]
        losses[prefix + 'uncert'] = self.use_confidence * self.safely_reduce_losses(cube_uncert.clone())
        if self.loss_w_dims > 0:
            losses[prefix + 'loss_dims'] = self.safely_reduce_losses(loss_dims) * self.loss_w_dims * self.loss_w_3d
        losses[prefix + 'loss_xy'] = self.safely_reduce_losses(loss_xy) * self.loss_w_xy * self.loss_w_3d
        losses[prefix + 'loss_z'] = self.safely_reduce_losses(loss_z) * self.loss_w_z * self.loss_w_3d
        losses[prefix + 'loss_pose'] = self.safely_reduce_losses(loss_pose) * self.loss_w_pose * self.loss_w_3d
        if self.loss_w_joint > 0:
            lj = torch.where(loss_joint < np.inf, loss_joint, torch.full_like(loss_joint, float('inf')))
            losses[prefix + 'loss_joint'] = self.safely_reduce_losses(lj, absent_if_none=True) * self.loss_w_joint * self.loss_w_3d
        with torch.no_grad():
            cube_3D = torch.cat((cube_x3d_y3d_z, cube_dims, cube_xy * im_ratios_per_box.unsqueeze(1),
                                 torch.exp(-cube_uncert).unsqueeze(1)), dim=1)
            verts = util.get_cuboid_verts_faces(cube_3D[:, :6], cube_pose)[0]
            pred_instances = [Instances(image_size) for image_size in im_current_dims]
            for c3, cp, vv, inst, cls_i, pb in zip(cube_3D.split(num_boxes_per_image), cube_pose.split(num_boxes_per_image),
                                                   verts.split(num_boxes_per_image), pred_instances,
                                                   box_classes.split(num_boxes_per_image), pred_boxes):
                inst.scores = c3[:, -1]
                inst.pred_classes = cls_i
                inst.pred_boxes = pb
                inst.pred_bbox3D = vv
                inst.pred_center_cam = c3[:, :3]
                inst.pred_center_2D = c3[:, 6:8]
                inst.pred_dimensions = c3[:, 3:6]
                inst.pred_pose = cp
        return pred_instances, losses

    def safely_reduce_losses(self, loss, absent_if_none=False):
        """roi_heads.py:2843-2851: mean over the finite entries; with none, `loss.mean()*0.0` exactly like the
        reference (NaN for NaN/Inf input, which its divergence guard then catches).  Branch-free on the device.
        absent_if_none: the reference drops the joint loss entirely in that case (roi_heads.py:2676) -> 0."""
        valid = (~(loss.isinf())) & (~(loss.isnan()))
        cnt = valid.sum()
        s = torch.where(valid, loss, torch.zeros_like(loss)).sum()
        none = s * 0.0 if absent_if_none else loss.mean() * 0.0
        return torch.where(cnt > 0, s / cnt.clamp(min=1), none)
