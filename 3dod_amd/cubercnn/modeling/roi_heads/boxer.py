"""ROIHeads_Boxer -- the 1000-cube proposal-and-scoring head, cubercnn/modeling/roi_heads/roi_heads.py:79-660 of
the reference: the AP path (use_pred_boxes / GT boxes -> ground normal -> propose -> score -> argmax, :492-505,647-660),
the MABO diagnostics on GT boxes (`experiment_type['output_recall_scores']`, :506-646: IoU3D of every proposal with the
ground-truth cube, the seven score functions, `accumulate_scores`, the 26 score combinations, the offset statistics) and the
two pseudo-ground-truth modes of training (`experiment_type['pseudo_gt']` = 'learn' | 'pseudo', :134-138,456-490).

Differences that are design, not semantics: all objects of all images of the batch are scored in ONE
cr_cubes_project_score launch with a per-object K (the reference handles batch size 1 and loops per object,
roi_heads.py:335,430,494-505); the ground plane is fitted with cr_ransac_plane (the reference calls pyransac3d on
the CPU, roi_heads.py:374-377).  The SAM-HQ mask model is external: masks come in through
batched_inputs[i]["masks"] (N_i,H,W) and become the 4-point rectangles of score_corners on the host; without masks
the reference's no-contour fallback is used (scorefunction.py:69-75)."""
from dataclasses import dataclass
from typing import Any, List

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


@dataclass
class Plotinfo:
    """roi_heads.py:50-60: what tools/eval_boxes.py plots from one image (`gt_cube_meshes` holds the (8,3) corners of each
    ground-truth cube here -- the reference stores pytorch3d meshes of the same corners)"""
    pred_cubes: Any
    gt_cube_meshes: List
    gt_boxes3D: Any
    gt_boxes: Any
    pred_boxes: Any
    gt_box_classes: Any
    mask_per_image: Any
    K: Any


# the 26 products of roi_heads.py:544-569, in the reference's order (entry 10 repeats entry 6 there, too);
# i = IoU2D, s = modified segment score, d = dimensions, c = corners, p = point cloud
COMBINATIONS = ("is", "id", "ic", "ip", "isd", "isc", "isp", "idc", "idp", "icp", "isp", "isdp", "iscp", "idcp", "isdcp",
                "sd", "sc", "sp", "sdc", "sdp", "scp", "sdcp", "dc", "dp", "dcp", "cp")


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

    def accumulate_scores(self, scores, IoU3D):
        """roi_heads.py:277-281: the IoU3D of the proposals ranked by `scores` (best first), as a running maximum -- entry k
        is the best IoU3D among the k+1 highest-scoring proposals.  numpy on the host like the reference (np.argsort's
        tie order is part of the MABO numbers)."""
        idx = np.argsort(scores)[::-1]
        return np.maximum.accumulate(np.array([IoU3D[i] for i in idx]))

    def forward(self, images, features, proposals, depth_maps, ground_maps, Ks, im_scales_ratio, masks=None,
                use_pred_boxes=True, generator=None, proposal_function='propose', experiment_type=None):
        """roi_heads.py:130-206.  eval: the AP path, or the MABO diagnostics when experiment_type['output_recall_scores'];
        training: pseudo-ground-truth generation on GT boxes (experiment_type['pseudo_gt'] = 'learn' | 'pseudo')."""
        ex = dict(experiment_type or {})
        if self.training:
            ex['use_pred_boxes'] = False                                       # :136
            return self._forward_gt_modes(images.image_sizes, proposals, depth_maps, ground_maps, Ks, im_scales_ratio, masks,
                                          generator, proposal_function, ex), {}
        if ex.get('output_recall_scores', False):
            assert not use_pred_boxes, "MABO compares with ground-truth cubes: use_pred_boxes must be False"
            return self._forward_gt_modes(images.image_sizes, proposals, depth_maps, ground_maps, Ks, im_scales_ratio, masks,
                                          generator, proposal_function, ex), {}
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
        # on GT boxes the reference scores IoU2D against the PROJECTED ground-truth cube (roi_heads.py:508,530), the other
        # terms against the annotated 2D box; instances without 3D ground truth fall back to the 2D box for both
        gt3d = None
        if not use_pred_boxes and all(i.has("gt_boxes3D") and i.has("gt_poses") for i in instances):
            # (one concatenation over the batch, not one per image: 64 small cat launches and their host time per step)
            b3 = torch.cat([i.gt_boxes3D for i in instances])
            gt3d = [torch.cat((b3[:, 6:], b3[:, 3:6], torch.cat([i.gt_poses for i in instances]).reshape(-1, 9)), 1)]
        return self._forward_cube(images.image_sizes, boxes, classes, depth_maps, ground_maps, Ks, im_scales_ratio,
                                  masks, generator, proposal_function, gt_cubes=gt3d), {}

    def _forward_box(self, features, proposals):
        feats = [features[f] for f in self.box_in_features]
        box_features = self.box_head(self.box_pooler(feats, [x.proposal_boxes for x in proposals]))
        pred_instances, _ = self.box_predictor.inference(self.box_predictor(box_features), proposals)
        return pred_instances

    @torch.no_grad()
    def _forward_cube(self, image_sizes, boxes: List[Boxes], classes, depth_maps, ground_maps, Ks, im_scales_ratio,
                      masks=None, generator=None, proposal_function='propose', gt_cubes=None):
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

        normals = self._ground_normals(depth_maps, ground_maps, K_img, generator)  # (B,3)

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
        iou_ref = None
        if gt_cubes is not None:                                               # 2D boxes of the projected ground-truth cubes
            g = torch.cat(gt_cubes).to(dev)[:, None, :].contiguous()
            one3 = torch.ones((g.shape[0], 3), device=dev)
            iou_ref = geo.cubes_project_score(g, K_obj, (W, H), ref, one3, one3, None, want=("boxes",))["boxes"][:, 0].contiguous()
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
            res = geo.cubes_project_score(cubes_t, K_obj, (W, H), ref, mu, sg, rects, want=(), iou_boxes=iou_ref)
            idx = res["argmax"]
            best = cubes_t[torch.arange(cubes_t.shape[0], device=dev), idx]    # (Ntot,15)
            verts = geo.cuboid_corners(best[:, :6].contiguous(), best[:, 6:].reshape(-1, 3, 3).contiguous())
            ctr_cam, dims, pose = best[:, :3], best[:, 3:6], best[:, 6:].reshape(-1, 3, 3)
            # one split per field (six calls) instead of six slices per image; the pieces of one image have equal lengths
            parts = [t.split(counts) for t in (res["best"], verts, ctr_cam, dims, pose, centers)]
            for i, (b, cls) in enumerate(zip(boxes, classes)):
                out_instances[i] = Instances._from_fields(image_sizes[i], dict(
                    pred_boxes=b, scores=parts[0][i], pred_classes=cls, pred_bbox3D=parts[1][i], pred_center_cam=parts[2][i],
                    pred_dimensions=parts[3][i], pred_pose=parts[4][i], pred_center_2D=parts[5][i]))
            if int(exhausted.item()) == 0:
                return out_instances
            PN.note_exhausted()

    def _ground_normals(self, depth_maps, ground_maps, K_img, generator=None):
        """roi_heads.py:345-428: ground normal of every image -- RANSAC (cr_ransac_plane_batched) over its strided ground
        pixels, or over all points if it has fewer than three, then the axis fix-ups -> (B,3)"""
        B, dev = depth_maps.shape[0], depth_maps.device
        pts = depth_to_points(depth_maps, K_img).reshape(B, -1, 3)
        eligible = None
        if ground_maps is not None:
            eligible = (ground_maps[:, ::5, ::5] > 0).reshape(B, -1)
            eligible = eligible | (eligible.sum(1, keepdim=True) < 3)
        triples = Plane.sample_triples_batched(eligible, B, pts.shape[1], 1000, dev, generator)
        neg_eq, _, _ = geo.ransac_plane_batched(pts, triples, eligible, thresh=0.05)
        return fix_ground_normal(-neg_eq[:, :3].t()).t().contiguous()

    # ------------------------------------------------------------------ MABO diagnostics / pseudo ground truth (GT boxes)
    @torch.no_grad()
    def _forward_gt_modes(self, image_sizes, instances, depth_maps, ground_maps, Ks, im_scales_ratio, masks, generator,
                          proposal_function, ex):
        """the branches of roi_heads.py:304-646 that work on ground-truth boxes.  Like the reference they handle ONE image per
        call (Ks[0], the masks of image 0: :335,430); a batch is walked image by image and a list of per-image results comes
        back when it holds more than one image."""
        out = []
        for i, inst in enumerate(instances):
            m = None if masks is None else masks[i]
            g = None if ground_maps is None else ground_maps[i:i + 1]
            out.append(self._gt_modes_one_image(image_sizes[i], inst, depth_maps[i:i + 1], g, Ks[i], im_scales_ratio[i], m,
                                                generator, proposal_function, ex))
        return out[0] if len(out) == 1 else out

    def _gt_modes_one_image(self, image_size, inst, depth_map, ground_map, K, ratio, masks, generator, proposal_function, ex):
        from ....ProposalNetwork.scoring import scorefunction as SF
        from ....ProposalNetwork.utils.utils import iou_3d
        from ....ProposalNetwork.utils.conversions import cubes_to_box
        from ...util import math_util as util
        dev = depth_map.device
        P = self.number_of_proposals
        gt_boxes, gt_classes = inst.gt_boxes, inst.gt_classes
        n_gt = len(gt_boxes)
        if n_gt == 0:                                                          # :330-333
            return [inst] if not self.training else ([inst], {})
        H, W = image_size
        im_shape = (W, H)
        K_img = (torch.as_tensor(K, dtype=torch.float32) / ratio).clone()
        K_img[-1, -1] = 1
        K_img = K_img.to(dev)
        prior = self.priors_dims_per_cat.detach()[0][gt_classes]
        mu, sg = prior[:, 0, :].contiguous(), prior[:, 1, :].contiguous()
        normal = self._ground_normals(depth_map, ground_map, K_img[None], generator)[0]
        gt3d, gt_poses = inst.gt_boxes3D, inst.gt_poses
        gt_cubes = Cubes(torch.cat((gt3d[:, 6:], gt3d[:, 3:6], gt_poses.reshape(n_gt, 9)), 1)[:, None, :])     # (n,1,15)
        ref = gt_boxes.tensor.contiguous()
        rects = None if masks is None else geo.mask_rects([masks.to(dev)])[0]     # NaN row = empty mask -> the no-contour fallback

        if isinstance(proposal_function, (list, tuple)):                       # :510-519: IoU3D of several samplers at once
            IoU3Ds = torch.zeros((n_gt, len(proposal_function), P), device=dev)
            for k, fn in enumerate(proposal_function):
                cubes, _, _, _ = self.predict_cubes(gt_boxes, (mu, sg), depth_map, im_shape, K_img, fn, normal, gt_cubes, generator)
                for j in range(n_gt):
                    IoU3Ds[j, k, :] = iou_3d(gt_cubes[j], cubes[j])
            return IoU3Ds

        pred_cubes, pred_boxes, stats_image, stats_ranges = self.predict_cubes(gt_boxes, (mu, sg), depth_map, im_shape, K_img,
                                                                               proposal_function, normal, gt_cubes, generator)
        cubes_t = pred_cubes.tensor.to(dev).contiguous()
        # IoU2D is taken against the PROJECTED ground-truth cube (:459,530), the aspect-ratio term of score_dimensions and
        # the corner score against the annotated 2D box and the object's mask (:460-461,537-540): `iou_boxes` of the scoring
        # kernel; the product is formed on the host in float32 like the reference's numpy product
        gt_proj = torch.cat([b.tensor for b in cubes_to_box(gt_cubes, K_img, im_shape)]).contiguous()
        K_obj = K_img[None].expand(n_gt, 3, 3).contiguous()

        if self.training and ex.get('pseudo_gt') == 'learn':                    # :456-460: every proposal with its IoU2D
            pred_cubes.scores = geo.cubes_project_score(cubes_t, K_obj, im_shape, ref, mu, sg, None, want=("iou",))["iou"]
            return pred_cubes
        b = geo.cubes_project_score(cubes_t, K_obj, im_shape, ref, mu, sg, rects, want=("corners", "iou", "dim", "corner"),
                                    iou_boxes=gt_proj)
        iou2d, dim, cor = b["iou"].cpu().numpy(), b["dim"].cpu().numpy(), b["corner"].cpu().numpy()
        combined = iou2d * dim * cor
        best = np.argmax(combined, axis=1)
        ar = torch.arange(n_gt, device=dev)
        best_t = torch.as_tensor(best, device=dev)
        out_cubes = Cubes(cubes_t[ar, best_t][:, None, :].clone(),
                          scores=torch.as_tensor(combined[np.arange(n_gt), best], device=dev)[:, None], labels=gt_classes)
        # (roi_heads.py:475,622 index the FIRST object's proposal boxes with every object's best index; kept as written)
        pred_boxes_out = Boxes(pred_boxes[0].tensor[best_t])

        def pack(boxes2d):
            r = Instances(image_size)
            r.pred_boxes = boxes2d
            r.scores = out_cubes.scores.squeeze(1)
            r.pred_classes = out_cubes.labels
            r.pred_bbox3D = out_cubes.get_all_corners().squeeze(1)
            r.pred_center_cam = out_cubes.centers.squeeze(1)
            r.pred_dimensions = out_cubes.dimensions.squeeze(1)
            r.pred_pose = out_cubes.rotations.squeeze(1)
            r.pred_center_2D = r.pred_boxes.get_centers()
            return r
        if self.training:
            if ex.get('pseudo_gt') != 'pseudo':
                raise ValueError("ROIHeads_Boxer in training mode generates pseudo ground truth: experiment_type['pseudo_gt'] "
                                 "must be 'learn' or 'pseudo'")
            return [pack(pred_boxes_out)]                                      # :462-490
        if not ex.get('output_recall_scores', False):
            return [pack(gt_boxes)]                                            # the AP packing on GT boxes (:647-660)

        # ---------------- MABO (:520-646)
        pts = depth_to_points(depth_map[0], K_img)                              # (h,w,3) at stride 5
        if ground_map is not None:
            pts = pts[ground_map[0, ::5, ::5] == 0]
        pts = pts.reshape(-1, 3)
        corners2d = b["corners"]                                               # (n,P,8,2)
        iou3d = np.stack([iou_3d(gt_cubes[i], pred_cubes[i]).cpu().numpy() for i in range(n_gt)])
        pc = np.stack([SF.score_point_cloud(pts, pred_cubes[i]).cpu().numpy() for i in range(n_gt)]).astype(np.float64)
        if masks is None:
            raise ValueError("the MABO segment scores need the objects' masks (batched_inputs[i]['masks'])")
        mk = masks.to(dev)
        seg = np.stack([SF.score_segmentation(mk[i].reshape(mk.shape[-2:]), corners2d[i:i + 1]).cpu().numpy() for i in range(n_gt)])
        segm = np.stack([SF.score_mod_segmentation(mk[i].reshape(mk.shape[-2:]), corners2d[i:i + 1]).cpu().numpy() for i in range(n_gt)])
        names = ("IoU2D", "seg", "dim", "combined", "random", "point_c", "seg_mod", "corner")
        scores = {k: np.zeros((n_gt, P)) for k in names}
        combinations = np.zeros((n_gt, len(COMBINATIONS)))
        stats_off = np.zeros((n_gt, 10))
        gt_meshes, empty = [], 0
        gt_corners = gt_cubes.get_all_corners()[:, 0]
        for i in range(n_gt):
            acc = lambda sc: self.accumulate_scores(sc, iou3d[i])
            term = {"i": iou2d[i], "s": segm[i], "d": dim[i], "c": cor[i], "p": pc[i]}
            rnd = np.random.rand(P)                                            # (the reference draws from numpy's global stream)
            for k, v in (("IoU2D", iou2d[i]), ("point_c", pc[i]), ("seg", seg[i]), ("dim", dim[i]), ("seg_mod", segm[i]),
                         ("corner", cor[i]), ("combined", combined[i]), ("random", rnd)):
                scores[k][i, :] = acc(v)
            for k, combo in enumerate(COMBINATIONS):
                prod = term[combo[0]]
                for ch in combo[1:]:
                    prod = prod * term[ch]
                combinations[i, k] = acc(prod)[0]
            gt_meshes.append(gt_corners[i])
            empty += int(np.count_nonzero(iou3d[i] == 0.0) / iou3d[i].size * 100)
            pc_i = out_cubes[i]
            rng = np.asarray(stats_ranges[i].cpu() if torch.is_tensor(stats_ranges) else stats_ranges[i], dtype=np.float64)
            off = [[iou3d[i].max()],
                   abs(gt_cubes[i].centers.cpu().numpy() - pc_i.centers.cpu().numpy())[0][0] / rng[:3],
                   abs(gt_cubes[i].dimensions.cpu().numpy() - pc_i.dimensions.cpu().numpy())[0][0] / rng[3:6],
                   abs(util.mat2euler(gt_cubes[i].rotations[0][0].cpu()) - util.mat2euler(pc_i.rotations[0][0].cpu())) / rng[6:]]
            stats_off[i] = [x for part in off for x in part]
        p_info = Plotinfo(out_cubes, gt_meshes, gt3d, gt_boxes, pred_boxes_out, gt_classes, masks, K_img.cpu().numpy())
        return (p_info, scores["IoU2D"], scores["seg"], scores["dim"], scores["combined"], scores["random"], scores["point_c"],
                empty / n_gt, stats_image, stats_off, scores["seg_mod"], scores["corner"], combinations)
