"""ROIHeads_Boxer -- the 1000-cube proposal-and-scoring head, cubercnn/modeling/roi_heads/roi_heads.py:79-660 of
the reference (AP path: use_pred_boxes / GT boxes -> ground normal -> propose -> score -> argmax).

Differences that are design, not semantics: all objects of all images of the batch are scored in ONE
cr_cubes_project_score launch with a per-object K (the reference handles batch size 1 and loops per object,
roi_heads.py:335,430,494-505); the ground plane is fitted with cr_ransac_plane (the reference calls pyransac3d on
the CPU, roi_heads.py:374-377).  The SAM-HQ mask model is external: masks come in through
batched_inputs[i]["masks"] (N_i,H,W) and become the 4-point rectangles of score_corners on the host; without masks
the reference's no-contour fallback is used (scorefunction.py:69-75)."""
from typing import List

import numpy as np
import torch
import torch.nn as nn

from ....d2lite import ROI_HEADS_REGISTRY, Boxes, Instances
from .... import geometry as geo
from ....ProposalNetwork.proposals import proposals as PN
from ....ProposalNetwork.utils.plane import Plane
from ....ProposalNetwork.utils.spaces import Cubes
from .fast_rcnn import FastRCNNOutputs, batched_nms
from .roi_heads import StandardROIHeads


def depth_to_points(depth, K, use_nth=5):
    """roi_heads.py:345-356, reproduced as written: strided pixel INDICES with the full-resolution intrinsics.
    depth (H,W) with K (3,3), or a batch (B,H,W) with K (B,3,3)."""
    dp = depth[..., ::use_nth, ::use_nth]
    Hs, Ws = dp.shape[-2:]
    v, u = torch.meshgrid(torch.arange(Hs, device=depth.device, dtype=torch.float32),
                          torch.arange(Ws, device=depth.device, dtype=torch.float32), indexing="ij")
    k = lambda r, c: K[..., r, c, None, None] if K.dim() == 3 else K[r, c]
    x = (u - k(0, 2)) * dp / k(0, 0)
    y = (v - k(1, 2)) * dp / k(1, 1)
    return torch.stack((x, y, dp), -1)


def fix_ground_normal(nv):
    """roi_heads.py:411-428 axis fix-ups, branch-free on the device."""
    n0, n1, n2 = nv[0], nv[1], nv[2]
    back = n2.abs() > n1.abs()
    a0, a1, a2 = torch.where(back, n0, n0), torch.where(back, n2, n1), torch.where(back, -n1, n2)
    side = a0.abs() > a1.abs()
    b0, b1, b2 = torch.where(side, -a2, a0), torch.where(side, a0, a1), torch.where(side, a1, a2)
    flip = b1 < 0
    out = torch.stack((b0, b1, b2))
    return torch.where(flip, -out, out)


@ROI_HEADS_REGISTRY.register()
class ROIHeads_Boxer(StandardROIHeads):
    def __init__(self, cfg, input_shape, priors=None):
        ret = StandardROIHeads.from_config(cfg, input_shape)
        ret["box_predictor"] = FastRCNNOutputs(cfg, ret["box_head"].output_shape)
        super().__init__(**ret)
        self.dims_priors_enabled = cfg.MODEL.ROI_CUBE_HEAD.DIMS_PRIORS_ENABLED
        self.number_of_proposals = cfg.MODEL.ROI_CUBE_HEAD.NUMBER_OF_PROPOSALS
        if self.dims_priors_enabled and priors is not None:
            self.priors_dims_per_cat = nn.Parameter(torch.FloatTensor(priors['priors_dims_per_cat']).unsqueeze(0))
        else:
            self.priors_dims_per_cat = nn.Parameter(torch.ones(1, self.num_classes, 2, 3))

    def predict_cubes(self, gt_boxes, priors, depth_maps_tensor, im_shape, K, proposal_function, normal_vec, gt_3d=None,
                      generator=None):
        """roi_heads.py:283-302: the proposals of ONE image by the sampler named `proposal_function` ('propose' = the
        method's own sampler on the fused kernel; 'random', 'xy', 'z', 'dim', 'rotation', 'aspect' = the ablation
        samplers of ProposalNetwork/proposals/proposals.py:20-336) -> (Cubes, list of Boxes, stats, ranges)"""
        from ....ProposalNetwork.utils.conversions import cubes_to_box
        from ....ProposalNetwork.utils.utils import Draws
        if proposal_function not in PN.PROPOSAL_FUNCTIONS:
            raise ValueError(f"unknown proposal function '{proposal_function}' (one of {sorted(PN.PROPOSAL_FUNCTIONS)})")
        ref = Boxes(gt_boxes) if torch.is_tensor(gt_boxes) else gt_boxes
        depth = depth_maps_tensor.squeeze()
        if proposal_function == 'propose':
            cubes, stats, ranges = PN.propose(ref, depth, priors, im_shape, K, number_of_proposals=self.number_of_proposals,
                                              gt_cubes=gt_3d, ground_normal=normal_vec, generator=generator)
        else:
            cubes, stats, ranges = PN.PROPOSAL_FUNCTIONS[proposal_function](
                ref, depth, priors, im_shape, K, number_of_proposals=self.number_of_proposals, gt_cubes=gt_3d,
                rng=Draws(generator))
        return cubes, cubes_to_box(cubes, K, im_shape), stats, ranges

    def forward(self, images, features, proposals, depth_maps, ground_maps, Ks, im_scales_ratio, masks=None,
                use_pred_boxes=True, generator=None, proposal_function='propose'):
        """eval-mode AP path (roi_heads.py:130-206)."""
        assert not self.training, "the pseudo-GT training modes of ROIHeads_Boxer are not built"
        if use_pred_boxes:
            pred = self._forward_box(features, proposals)
            instances = []
            for p in pred:
                keep = batched_nms(p.pred_boxes.tensor, p.scores, torch.zeros_like(p.pred_classes), 0.5)[:20]
                instances.append(p[keep])                                      # class-agnostic NMS, keep <= 20
            boxes = [i.pred_boxes for i in instances]
            classes = [i.pred_classes for i in instances]
        else:
            instances = proposals                                              # GT instances
            boxes = [i.gt_boxes for i in instances]
            classes = [i.gt_classes for i in instances]
        return self._forward_cube(images.image_sizes, boxes, classes, depth_maps, ground_maps, Ks, im_scales_ratio,
                                  masks, generator, proposal_function), {}

    def _forward_box(self, features, proposals):
        feats = [features[f] for f in self.box_in_features]
        box_features = self.box_head(self.box_pooler(feats, [x.proposal_boxes for x in proposals]))
        pred_instances, _ = self.box_predictor.inference(self.box_predictor(box_features), proposals)
        return pred_instances

    @torch.no_grad()
    def _forward_cube(self, image_sizes, boxes: List[Boxes], classes, depth_maps, ground_maps, Ks, im_scales_ratio,
                      masks=None, generator=None, proposal_function='propose'):
        """roi_heads.py:304-660 (use_pred_boxes branch :492-505 and the Instances packing :647-660).  The reference
        walks the images one by one; here every stage (ground-plane fit, proposals, mask rectangles, scoring) is one
        launch for the whole batch, with a single host sync (the rejection sampler's exhausted flag)."""
        dev = depth_maps.device
        P = self.number_of_proposals
        out_instances = [Instances(s) for s in image_sizes]
        counts = [len(b) for b in boxes]
        if sum(counts) == 0:
            return out_instances
        assert len({tuple(s) for s in image_sizes}) == 1, "one clamp window per launch: batch images of one size"
        H, W = image_sizes[0]
        B = len(boxes)
        K_img = torch.stack([torch.as_tensor(Ks[i], dtype=torch.float32) / im_scales_ratio[i] for i in range(B)])
        K_img[:, -1, -1] = 1
        K_img = K_img.to(dev)
        # numpy, not torch.repeat_interleave on the CPU: that opens an OpenMP region, whose spinning workers can eat a
        # container's CPU quota and stall this thread for the rest of the scheduler period (~90 ms measured)
        img_idx = torch.from_numpy(np.repeat(np.arange(B, dtype=np.int32), counts)).to(dev)
        ref = torch.cat([b.tensor for b in boxes]).contiguous()
        prior = self.priors_dims_per_cat.detach()[0][torch.cat(list(classes))]   # (Ntot,2,3)
        mu, sg = prior[:, 0, :].contiguous(), prior[:, 1, :].contiguous()

        # ground normal of every image: RANSAC over its (strided) ground pixels, or over all points if it has < 3
        pts = depth_to_points(depth_maps, K_img).reshape(B, -1, 3)
        eligible = None
        if ground_maps is not None:
            eligible = (ground_maps[:, ::5, ::5] > 0).reshape(B, -1)
            eligible = eligible | (eligible.sum(1, keepdim=True) < 3)
        triples = Plane.sample_triples_batched(eligible, B, pts.shape[1], 1000, dev, generator)
        neg_eq, _, _ = geo.ransac_plane_batched(pts, triples, eligible, thresh=0.05)
        normals = fix_ground_normal(-neg_eq[:, :3].t()).t().contiguous()        # (B,3)

        rects = None
        if masks is not None and any(m is not None for m in masks):
            if all(m is not None for m in masks):
                rects = geo.mask_rects([m.to(dev) for m in masks])[0]             # NaN row = empty mask -> fallback
            else:
                rects = torch.full((sum(counts), 4, 2), float("nan"), device=dev)
                have = torch.cat([torch.full((n,), m is not None) for n, m in zip(counts, masks)]).to(dev)
                rects[have] = geo.mask_rects([m.to(dev) for m in masks if m is not None])[0]
        K_obj = K_img[img_idx.long()].contiguous()
        centers = (ref[:, :2] + ref[:, 2:]) / 2                                 # Boxes.get_centers for every object
        while True:
            # proposals -> scores -> best cube -> Instances are all issued before the rejection sampler's flag is read, so
            # the host never waits in the middle of the batch and its packing overlaps the kernels; a non-zero flag (rare
            # once the round count has adapted) redoes them
            if proposal_function == 'propose':
                cubes_t, exhausted = PN.propose_batched(ref, img_idx, depth_maps, (mu, sg), K_img, P, normals,
                                                        generator=generator, defer_check=True)
            else:
                # an ablation sampler (roi_heads.py:286-297): image by image, as the reference does
                parts, o = [], 0
                for i, n in enumerate(counts):
                    if n:
                        c, _, _, _ = self.predict_cubes(boxes[i], (mu[o:o + n], sg[o:o + n]), depth_maps[i], (W, H), K_img[i],
                                                        proposal_function, normals[i], generator=generator)
                        parts.append(c.tensor.to(dev))
                    o += n
                cubes_t = torch.cat(parts).contiguous()
                exhausted = torch.zeros((1,), dtype=torch.int32, device=dev)
            res = geo.cubes_project_score(cubes_t, K_obj, (W, H), ref, mu, sg, rects, want=())
            idx = res["argmax"]
            best = cubes_t[torch.arange(cubes_t.shape[0], device=dev), idx]    # (Ntot,15)
            verts = geo.cuboid_corners(best[:, :6].contiguous(), best[:, 6:].reshape(-1, 3, 3).contiguous())
            ctr_cam, dims, pose = best[:, :3], best[:, 3:6], best[:, 6:].reshape(-1, 3, 3)
            off = 0
            for i, (b, cls) in enumerate(zip(boxes, classes)):
                n = counts[i]
                sl = slice(off, off + n)
                out_instances[i] = Instances(image_sizes[i], pred_boxes=b, scores=res["best"][sl], pred_classes=cls,
                                             pred_bbox3D=verts[sl], pred_center_cam=ctr_cam[sl], pred_dimensions=dims[sl],
                                             pred_pose=pose[sl], pred_center_2D=centers[sl])
                off += n
            if int(exhausted.item()) == 0:
                return out_instances
            PN.note_exhausted()
