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
    """roi_heads.py:345-356, reproduced as written: strided pixel INDICES with the full-resolution intrinsics."""
    dp = depth[::use_nth, ::use_nth]
    Hs, Ws = dp.shape
    v, u = torch.meshgrid(torch.arange(Hs, device=depth.device, dtype=torch.float32),
                          torch.arange(Ws, device=depth.device, dtype=torch.float32), indexing="ij")
    x = (u - K[0, 2]) * dp / K[0, 0]
    y = (v - K[1, 2]) * dp / K[1, 1]
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

    def forward(self, images, features, proposals, depth_maps, ground_maps, Ks, im_scales_ratio, masks=None,
                use_pred_boxes=True, generator=None):
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
                                  masks, generator), {}

    def _forward_box(self, features, proposals):
        feats = [features[f] for f in self.box_in_features]
        box_features = self.box_head(self.box_pooler(feats, [x.proposal_boxes for x in proposals]))
        pred_instances, _ = self.box_predictor.inference(self.box_predictor(box_features), proposals)
        return pred_instances

    @torch.no_grad()
    def _forward_cube(self, image_sizes, boxes: List[Boxes], classes, depth_maps, ground_maps, Ks, im_scales_ratio,
                      masks=None, generator=None):
        """roi_heads.py:304-660 (use_pred_boxes branch :492-505 and the Instances packing :647-660)."""
        dev = depth_maps.device
        P = self.number_of_proposals
        cubes_all, K_all, ref_all, mu_all, sg_all, rect_all, have_rect = [], [], [], [], [], [], False
        for i, (b, cls) in enumerate(zip(boxes, classes)):
            n = len(b)
            if n == 0:
                continue
            K = (torch.as_tensor(Ks[i], dtype=torch.float32) / im_scales_ratio[i]).to(dev)
            K[-1, -1] = 1
            prior = self.priors_dims_per_cat.detach()[0][cls]                  # (n,2,3)
            mu, sg = prior[:, 0, :].contiguous(), prior[:, 1, :].contiguous()
            pts = depth_to_points(depth_maps[i], K)
            if ground_maps is not None:
                g = ground_maps[i][::5, ::5] > 0
                gp = pts[g]
                gp = gp if gp.shape[0] >= 3 else pts.reshape(-1, 3)
            else:
                gp = pts.reshape(-1, 3)
            neg_eq, _ = Plane().fit_parallel(gp.contiguous(), thresh=0.05, maxIteration=1000, generator=generator)
            normal = fix_ground_normal(-neg_eq[:3])
            H, W = image_sizes[i]
            cubes, _, _ = PN.propose(b, depth_maps[i], (mu, sg), (W, H), K, P, ground_normal=normal, generator=generator)
            cubes_all.append(cubes.tensor)
            K_all.append(K.unsqueeze(0).expand(n, 3, 3))
            ref_all.append(b.tensor)
            mu_all.append(mu)
            sg_all.append(sg)
            if masks is not None and masks[i] is not None:
                have_rect = True
                rect_all.append(geo.mask_rects(masks[i].to(dev))[0])           # NaN row = empty mask -> fallback rect
            else:
                rect_all.append(torch.full((n, 4, 2), float("nan"), device=dev))
        out_instances = [Instances(s) for s in image_sizes]
        if not cubes_all:
            return out_instances
        sizes = {tuple(s) for s in image_sizes}
        assert len(sizes) == 1, "one clamp window per launch: batch images of one size"
        H, W = image_sizes[0]
        cubes_t = torch.cat(cubes_all)
        res = geo.cubes_project_score(cubes_t, torch.cat(K_all).contiguous(), (W, H), torch.cat(ref_all).contiguous(),
                                      torch.cat(mu_all), torch.cat(sg_all),
                                      torch.cat(rect_all) if have_rect else None, want=())
        idx = res["argmax"]
        best = cubes_t[torch.arange(cubes_t.shape[0], device=dev), idx]        # (Ntot,15)
        verts = geo.cuboid_corners(best[:, :6].contiguous(), best[:, 6:].reshape(-1, 3, 3).contiguous())
        off = 0
        for i, (b, cls) in enumerate(zip(boxes, classes)):
            n = len(b)
            inst = out_instances[i]
            sl = slice(off, off + n)
            inst.pred_boxes = b
            inst.scores = res["best"][sl]
            inst.pred_classes = cls
            inst.pred_bbox3D = verts[sl]
            inst.pred_center_cam = best[sl, :3]
            inst.pred_dimensions = best[sl, 3:6]
            inst.pred_pose = best[sl, 6:].reshape(-1, 3, 3)
            inst.pred_center_2D = b.get_centers()
            off += n
        return out_instances
