"""ROIHeads3DScore -- the weakly supervised 3D head (reference: cubercnn/modeling/roi_heads/roi_heads.py:664-1946,
configs/Omni_combined.yaml): the box branch and the cube head are those of ROIHeads3D, but the cube branch is trained
from 2D boxes, a metric depth map and a ground mask instead of 3D labels:

    iou                 GIoU between the GT 2D box and the hull of the projected cuboid
    pose_alignment      cuboids of one image should share their orientation
    pose_ground(2)      the cuboid's up-axis (or whole frame) should match the RANSAC ground normal
    z / z_pseudo_gt_*   depth from the 2D box area / from the depth map under the box (median) or its centre
    dims                hinge on the class prior of (w, h, l)

All of them are batched tensor expressions over the foreground RoIs (weak_losses.py); the per-box depth median and the
plane fit are HIP kernels.
    segmentation        hull of the projected corners as a soft polygon mask vs the object's mask (focal loss; one fused kernel)
    depth               depth extent of the cuboid vs the 10..90 % depth range under the object's mask
The last two need a mask per object: the reference runs SAM-HQ on the GT boxes (roi_heads.py:881-883); that model is out
of scope, so the masks come from a pluggable `segmentor` callable.
"""
from typing import Dict

import torch

from ....d2lite import ROI_HEADS_REGISTRY, Boxes, Instances, ShapeSpec, get_event_storage
from ...util import math_util as util
from . import weak_losses as W
from .roi_heads import ROIHeads3D, select_foreground_proposals

SQRT_2_CONSTANT = 1.4142135623730951
POSSIBLE_LOSSES = ['dims', 'pose_alignment', 'pose_ground', 'pose_ground2', 'iou', 'segmentation', 'z', 'z_pseudo_gt_patch',
                   'z_pseudo_gt_center', 'depth']


@ROI_HEADS_REGISTRY.register()
class ROIHeads3DScore(ROIHeads3D):
    def __init__(self, cfg, input_shape: Dict[str, ShapeSpec], priors=None):
        super().__init__(cfg, input_shape, priors=priors)
        c = cfg.MODEL.ROI_CUBE_HEAD
        self.loss_w_iou = c.LOSS_W_IOU
        self.loss_w_seg = c.LOSS_W_SEG
        self.loss_w_normal_vec = c.LOSS_W_NORMAL_VEC
        self.loss_w_depth = c.LOSS_W_DEPTH
        self.loss_functions = list(cfg.loss_functions)
        assert all(x in POSSIBLE_LOSSES for x in self.loss_functions), \
            f'loss functions must be in {POSSIBLE_LOSSES}, but was {self.loss_functions}'
        # 'segmentation' / 'depth' need one mask per ground-truth object.  The reference gets them from SAM-HQ prompted with
        # the GT boxes (object_masks, roi_heads.py:1000-1016; init_segmentation); that model is not built,
        # so the masks come from a pluggable callable: segmentor(images_raw (B,3,H,W) uint8 RGB, targets) -> list per
        # image of (N_i,1,H,W) bool tensors.  Without one these two losses raise at the first training step.
        self.segmentor = None
        self._hull_fn = None
        self._focal_fn = None
        # test hooks: CPU restatements (oracle/weak.py) of the two kernels; None = the HIP path, which refuses CPU tensors
        self._median_fn = None
        self._plane_cls = None
        self._ransac_triples = None

    # ------------------------------------------------------------------ forward (roi_heads.py:854-914)
    def forward(self, images, images_raw, ground_maps, depth_maps, features, proposals, Ks, im_scales_ratio, targets=None):
        im_dims = [tuple(s) for s in images.image_sizes]
        if self.training:
            proposals = self.label_and_sample_proposals(proposals, targets)
            losses = self._forward_box(features, proposals)
            if self.loss_w_3d > 0:
                masks, keys = self.object_masks(images_raw, targets)
                instances_3d, losses_cube = self._forward_cube(features, proposals, Ks, im_dims, im_scales_ratio, masks, keys,
                                                               ground_maps, depth_maps)
                losses.update(losses_cube)
            else:
                instances_3d = None
            return instances_3d, losses
        return ROIHeads3D.forward(self, images, features, proposals, Ks, im_scales_ratio, targets)

    @property
    def needs_masks(self):
        return 'segmentation' in self.loss_functions or 'depth' in self.loss_functions

    def object_masks(self, images_raw, targets):
        """roi_heads.py:864-885,1000-1016: (masks (Nm,H,W) uint8 over all GT objects of the batch in target order, the
        gt_boxes3D[:, 0] value of each -- the key the reference looks masks up by) or (None, None)."""
        if not self.needs_masks:
            return None, None
        if self.segmentor is None:
            raise RuntimeError("cfg.loss_functions asks for 'segmentation' / 'depth' but no segmentor is set: assign a callable "
                               "(images_raw, targets) -> per-image (N,1,H,W) masks to model.roi_heads.segmentor")
        raw = images_raw.tensor if hasattr(images_raw, "tensor") else images_raw
        per_image = self.segmentor(raw, targets)
        masks = torch.cat([m.reshape(-1, *m.shape[-2:]) for m in per_image]).to(torch.uint8).contiguous()
        keys = torch.cat([t.gt_boxes3D[:, 0] for t in targets])
        assert masks.shape[0] == keys.shape[0], "one mask per ground-truth object"
        return masks, keys

    # ------------------------------------------------------------------ cube branch (roi_heads.py:1319-1820)
    def _forward_cube(self, features, instances, Ks, im_current_dims, im_scales_ratio, masks_all_images=None,
                      first_occurrence_indices=None, ground_maps=None, depth_maps=None):
        if not self.training:
            # the decode of the two heads is the same code (roi_heads.py:1351-1501 == 2271-2436)
            return ROIHeads3D._forward_cube(self, features, instances, Ks, im_current_dims, im_scales_ratio)
        feats = [features[f] for f in self.in_features]
        losses = {}
        self.normalize_factor = max(sum([i.gt_classes.numel() for i in instances]), 1.0)
        proposals, _ = select_foreground_proposals(instances, self.num_classes)
        proposal_boxes = [x.proposal_boxes for x in proposals]
        pred_boxes = [x.pred_boxes for x in proposals]
        box_classes = torch.cat([p.gt_classes for p in proposals], dim=0) if len(proposals) else torch.empty(0)
        gt_boxes3D = torch.cat([p.gt_boxes3D for p in proposals], dim=0)
        gt_poses = torch.cat([p.gt_poses for p in proposals], dim=0)
        assert len(gt_poses) == len(gt_boxes3D) == len(box_classes)

        proposal_boxes_scaled = self.scale_proposals(proposal_boxes)
        n = sum(len(b) for b in proposal_boxes_scaled)
        if n == 0:
            return instances, {}
        cube_features = self.cube_pooler(feats, proposal_boxes_scaled).flatten(1)
        num_boxes_per_image = [len(i) for i in proposals]
        masks, mask_keys = masks_all_images, first_occurrence_indices
        if isinstance(masks, (list, tuple)):                      # the reference's form: list of (1,H,W) masks + {key: index}
            masks = torch.cat([m.reshape(-1, *m.shape[-2:]) for m in masks]).to(torch.uint8).contiguous()
            mask_keys = torch.tensor([k for k, _ in sorted(mask_keys.items(), key=lambda kv: kv[1])], dtype=torch.float32,
                                     device=masks.device)
        losses, cube_3D, cube_pose = self.weak_losses_flat(
            cube_features, box_classes, torch.cat([b.tensor for b in proposal_boxes], dim=0),
            torch.cat([x.gt_boxes.tensor for x in proposals]), gt_boxes3D, gt_poses, num_boxes_per_image, Ks,
            im_current_dims, im_scales_ratio, ground_maps, depth_maps, masks, mask_keys)

        # ---- packing of the decoded cuboids (roi_heads.py:1763-1815)
        pred_instances = [Instances(image_size) for image_size in im_current_dims]
        for cube_3D_i, cube_pose_i, inst, cls_i, boxes_i in zip(cube_3D.split(num_boxes_per_image),
                                                                cube_pose.split(num_boxes_per_image), pred_instances,
                                                                box_classes.split(num_boxes_per_image), pred_boxes):
            inst.scores = cube_3D_i[:, -1]
            inst.pred_classes = cls_i
            inst.pred_boxes = boxes_i
            inst.pred_bbox3D = util.get_cuboid_verts_faces(cube_3D_i[:, :6], cube_pose_i)[0]
            inst.pred_center_cam = cube_3D_i[:, :3]
            inst.pred_center_2D = cube_3D_i[:, 6:8]
            inst.pred_dimensions = cube_3D_i[:, 3:6]
            inst.pred_pose = cube_pose_i
        return pred_instances, losses

    def weak_losses_flat(self, cube_features, box_classes, src_boxes, gt_boxes, gt_boxes3D, gt_poses, num_boxes_per_image,
                         Ks, im_current_dims, im_scales_ratio, ground_maps, depth_maps, masks=None, mask_keys=None):
        """decode + the weak losses on a flat, image-major list of n foreground RoIs (roi_heads.py:1366-1760).
        cube_features (n, C*7*7); box_classes (n); src_boxes / gt_boxes (n,4); gt_boxes3D (n,9); gt_poses (n,3,3);
        num_boxes_per_image: host ints.  Returns (losses, cube_3D (n,9), cube_pose (n,3,3))."""
        losses = {}
        n = cube_features.shape[0]
        device = cube_features.device

        # every per-image constant the branch needs goes to the device in ONE small copy and is gathered per box:
        # K / ratio (9), fy of the original K, ratio, image height, clamp bounds of the projection (4), ground confidence
        B = len(num_boxes_per_image)
        rows = []
        for i in range(B):
            k = torch.as_tensor(Ks[i], dtype=torch.float32) / im_scales_ratio[i]
            k[-1, -1] = 1
            d = im_current_dims[i]
            gconf = 1.0
            if ground_maps is not None and tuple(ground_maps.image_sizes[i]) == (1, 1):
                gconf = 0.1
            rows.append(k.flatten().tolist() + [float(torch.as_tensor(Ks[i])[1][1]), float(im_scales_ratio[i]), float(d[0])]
                        + [float(v) for v in W._int_clamp_bounds(d[0]) + W._int_clamp_bounds(d[1])] + [gconf]
                        + [float(int(d[0] - 1)), float(int(d[1] - 1))])
        table = torch.tensor(rows, dtype=torch.float32)
        img_host = torch.tensor([i for i, num in enumerate(num_boxes_per_image) for _ in range(num)], dtype=torch.int64)
        if device.type == "cuda":
            table, img_host = table.pin_memory(), img_host.pin_memory()
        table, img = table.to(device, non_blocking=True), img_host.to(device, non_blocking=True)
        per_box = table[img]
        Ks_scaled_per_box = per_box[:, :9].reshape(n, 3, 3)
        focal_lengths_per_box, im_ratios_per_box, im_scales_per_box = per_box[:, 9], per_box[:, 10], per_box[:, 11]
        clamp_bounds = per_box[:, 12:16]
        im_scales_original_per_box = im_scales_per_box * im_ratios_per_box
        if self.virtual_depth:
            virtual_to_real = util.compute_virtual_scale_from_focal_spaces(
                focal_lengths_per_box, im_scales_original_per_box, self.virtual_focal, im_scales_per_box)
        else:
            virtual_to_real = 1.0

        src_widths = src_boxes[:, 2] - src_boxes[:, 0]
        src_heights = src_boxes[:, 3] - src_boxes[:, 1]
        src_ctr_x = src_boxes[:, 0] + 0.5 * src_widths
        src_ctr_y = src_boxes[:, 1] + 0.5 * src_heights

        # the number of foreground RoIs changes every step and hipBLASLt picks a kernel per problem size (~100 us of host
        # time for each unseen shape, x ~24 GEMMs fwd + bwd): the head runs on the rows padded to a multiple of 128, so only
        # a couple of shapes ever occur; the padding rows are sliced off (their gradient is zero)
        rows = -(-n // 128) * 128
        head_in = torch.nn.functional.pad(cube_features, (0, 0, 0, rows - n)) if rows != n else cube_features
        cube_2d_deltas, cube_z, cube_dims, cube_pose, cube_uncert = (
            t[:n] if t is not None else None for t in self.cube_head(head_in))
        fg_inds = torch.arange(n, device=device)
        cube_z = util.cluster_depth(cube_z, box_classes, src_boxes, getattr(self, "priors_z_scales", None),
                                    getattr(self, "z_type", "direct"), getattr(self, "priors_z_stats", None)).unsqueeze(1)
        cube_dims = cube_dims[fg_inds, box_classes, :]
        cube_pose = cube_pose[fg_inds, box_classes, :, :]
        if self.use_confidence:
            cube_uncert = cube_uncert[fg_inds, box_classes]
        cube_2d_deltas = cube_2d_deltas[fg_inds, box_classes, :]

        cube_x = src_ctr_x + src_widths * cube_2d_deltas[:, 0]
        cube_y = src_ctr_y + src_heights * cube_2d_deltas[:, 1]
        cube_xy = torch.cat((cube_x.unsqueeze(1), cube_y.unsqueeze(1)), dim=1)
        cube_dims_norm = cube_dims
        prior_dims_mean = prior_dims_std = None
        if self.dims_priors_enabled:
            prior_dims = self.priors_dims_per_cat.detach()[0][box_classes]
            prior_dims_mean, prior_dims_std = prior_dims[:, 0, :], prior_dims[:, 1, :]
            if self.dims_priors_func == 'sigmoid':
                cube_dims = util.scaled_sigmoid(cube_dims_norm, min=(prior_dims_mean - 3 * prior_dims_std).clip(0.0),
                                                max=(prior_dims_mean + 3 * prior_dims_std))
            elif self.dims_priors_func == 'exp':
                cube_dims = torch.exp(cube_dims_norm.clip(max=5)) * prior_dims_mean
        else:
            cube_dims = torch.exp(cube_dims_norm.clip(max=5))
        if self.allocentric_pose:
            cube_pose = util.R_from_allocentric(Ks_scaled_per_box, cube_pose, u=cube_x.detach(), v=cube_y.detach())
        cube_z = cube_z.squeeze(1)
        if self.virtual_depth:
            cube_z = cube_z * virtual_to_real

        prefix = 'Cube/'
        storage = get_event_storage()
        K = Ks_scaled_per_box
        gt_2d, gt_z, gt_dims = gt_boxes3D[:, :2], gt_boxes3D[:, 2], gt_boxes3D[:, 3:6]

        # predicted cuboids in metres and their 2D hulls (Cubes + cubes_to_box per RoI in the reference, :1547-1583)
        cube_x3d = cube_z * (cube_x - K[:, 0, 2]) / K[:, 0, 0]
        cube_y3d = cube_z * (cube_y - K[:, 1, 2]) / K[:, 1, 1]
        cubes = torch.cat((cube_x3d.unsqueeze(1), cube_y3d.unsqueeze(1), cube_z.unsqueeze(1), cube_dims,
                           cube_pose.reshape(n, 9)), dim=1)
        corners2d = W.project_cubes_to_corners(cubes.unsqueeze(1), K, clamp_bounds)
        proj_boxes = W.corners_to_boxes(corners2d)[:, 0]
        loss_iou = loss_pose = loss_z = loss_dims_w = loss_dims_h = loss_dims_l = None
        loss_pseudo_gt_z = loss_ground_rot = loss_seg = loss_depth = None
        if masks is not None and self.needs_masks:
            mask_idx = W.mask_index_of(gt_boxes3D[:, 0], mask_keys.to(gt_boxes3D.dtype))
            if 'segmentation' in self.loss_functions:
                # corners clamped into the image (x with dims[0]-1, y with dims[1]-1, as in the reference :1572-1575)
                bube = torch.stack((torch.minimum(corners2d[:, 0, :, 0].clamp(min=0), per_box[:, 17:18]),
                                    torch.minimum(corners2d[:, 0, :, 1].clamp(min=0), per_box[:, 18:19])), dim=-1)
                loss_seg = W.segment_loss(masks, bube, mask_idx, self._hull_fn, self._focal_fn)
            if 'depth' in self.loss_functions:
                corners_z = util.get_cuboid_verts_faces(cubes[:, :6], cube_pose)[0][..., 2]
                loss_depth = W.depth_range_loss(masks, mask_idx, depth_maps, corners_z, gt_boxes, img)
        if 'iou' in self.loss_functions:
            loss_iou = W.generalized_box_iou_loss(gt_boxes, proj_boxes, reduction='none').view(n, -1).mean(dim=1)
        if 'pose_alignment' in self.loss_functions:
            loss_pose = W.pose_alignment_loss(cube_pose, num_boxes_per_image)
        if loss_pose is not None:
            loss_pose = loss_pose.repeat(n)
        if 'pose_ground' in self.loss_functions or 'pose_ground2' in self.loss_functions:
            normals = W.ground_normals(ground_maps, depth_maps, K, id_samples=self._ransac_triples,
                                       plane_cls=self._plane_cls)
            normals = normals[img]
            valid_ground_maps_conf = per_box[:, 16]
            if 'pose_ground' in self.loss_functions:
                loss_ground_rot = 1 - torch.nn.functional.cosine_similarity(normals, cube_pose[:, 1, :], dim=1).abs()
                loss_ground_rot = loss_ground_rot * valid_ground_maps_conf
            if 'pose_ground2' in self.loss_functions:
                loss_ground_rot = 1 - W.so3_relative_angle(cube_pose, W.normal_to_rotation(normals), cos_angle=True)
                loss_ground_rot = loss_ground_rot * valid_ground_maps_conf
        if 'z_pseudo_gt_patch' in self.loss_functions:
            target = W.pseudo_gt_z_box(depth_maps, proj_boxes, img, median_fn=self._median_fn)
            loss_pseudo_gt_z = self.l1_loss(cube_z, target)
        elif 'z_pseudo_gt_center' in self.loss_functions:
            loss_pseudo_gt_z = self.l1_loss(cube_z, W.pseudo_gt_z_point(depth_maps, cube_xy, img))
        if 'z' in self.loss_functions:
            loss_z = W.z_search_loss(gt_boxes, cubes, K, clamp_bounds, proj_boxes)
        if 'dims' in self.loss_functions:
            loss_dims_w, loss_dims_h, loss_dims_l = W.dim_hinge_loss(prior_dims_mean, prior_dims_std, cube_dims)

        with torch.no_grad():
            total = 0
            for l, w in ((loss_iou, self.loss_w_iou), (loss_seg, self.loss_w_seg), (loss_depth, self.loss_w_depth),
                         (loss_pose, self.loss_w_pose), (loss_z, self.loss_w_z),
                         (loss_pseudo_gt_z, self.loss_w_z), (loss_dims_w, self.loss_w_dims), (loss_dims_h, self.loss_w_dims),
                         (loss_dims_l, self.loss_w_dims)):
                if l is not None:
                    total = total + l * w
            if loss_ground_rot is not None:      # weighted by the confidence a second time, as in the reference (:1663)
                total = total + loss_ground_rot * self.loss_w_normal_vec * valid_ground_maps_conf
            z_error = (cube_z - gt_z).abs()
            storage.put_scalar(prefix + 'z_error', z_error.mean(), smoothing_hint=False)
            storage.put_scalar(prefix + 'dims_error', (cube_dims - gt_dims).abs().mean(), smoothing_hint=False)
            storage.put_scalar(prefix + 'xy_error', (cube_xy - gt_2d).abs().mean(), smoothing_hint=False)
            storage.put_scalar(prefix + 'z_close', (z_error < 0.20).float().mean(), smoothing_hint=False)
            inter_wh = (torch.min(gt_boxes[:, 2:], proj_boxes[:, 2:]) - torch.max(gt_boxes[:, :2], proj_boxes[:, :2])).clamp(min=0)
            inter = inter_wh[:, 0] * inter_wh[:, 1]
            area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
            iou2d = torch.where(inter > 0, inter / (area(gt_boxes) + area(proj_boxes) - inter), torch.zeros_like(inter))
            storage.put_scalar(prefix + '2D IoU', iou2d.mean(), smoothing_hint=False)
            if not isinstance(total, int):
                storage.put_scalar(prefix + 'total_3D_loss', self.loss_w_3d * self.safely_reduce_losses(total),
                                   smoothing_hint=False)

        named = [('loss_iou', loss_iou, self.loss_w_iou), ('loss_pose', loss_pose, self.loss_w_pose),
                 ('loss_normal_vec', loss_ground_rot, self.loss_w_normal_vec), ('loss_seg', loss_seg, self.loss_w_seg),
                 ('loss_depth', loss_depth, self.loss_w_depth), ('loss_z', loss_z, self.loss_w_z),
                 ('loss_pseudo_gt_z', loss_pseudo_gt_z, self.loss_w_z), ('loss_dims_w', loss_dims_w, self.loss_w_dims),
                 ('loss_dims_h', loss_dims_h, self.loss_w_dims), ('loss_dims_l', loss_dims_l, self.loss_w_dims)]
        if self.use_confidence > 0:
            uncert_sf = SQRT_2_CONSTANT * torch.exp(-cube_uncert)
            named = [(k, l * uncert_sf if l is not None else None, w) for k, l, w in named]
            losses.update({prefix + 'uncert': self.use_confidence * self.safely_reduce_losses(cube_uncert.clone())})
            storage.put_scalar(prefix + 'conf', torch.exp(-cube_uncert.detach()).mean(), smoothing_hint=False)
        # safely_reduce_losses of every term in one masked reduction over the stacked (terms, n) matrix
        live = [(k, l, w) for k, l, w in named if l is not None]
        if live:
            L = torch.stack([l for _, l, _ in live])
            valid = torch.isfinite(L)
            cnt = valid.sum(1)
            tot = torch.where(valid, L, torch.zeros_like(L)).sum(1)
            red = torch.where(cnt > 0, tot / cnt.clamp(min=1), L.mean(1) * 0.0)
            wts = torch.tensor([w * self.loss_w_3d for _, _, w in live], dtype=red.dtype).to(red.device, non_blocking=True)
            red = red * wts
            for i, (k, _, _) in enumerate(live):
                losses[prefix + k] = red[i]

        cube_3D = torch.cat((torch.stack((cube_x3d, cube_y3d, cube_z)).T, cube_dims,
                             cube_xy * im_ratios_per_box.unsqueeze(1)), dim=1)
        if self.use_confidence:
            cube_3D = torch.cat((cube_3D, torch.exp(-cube_uncert).unsqueeze(1)), dim=1)
        return losses, cube_3D, cube_pose
