"""Golden vectors for ROIHeads_Boxer on ground-truth boxes: runs the REFERENCE's own ROIHeads_Boxer._forward_cube
(cubercnn/modeling/roi_heads/roi_heads.py:304-660) on CPU for

  mabo      eval mode, experiment_type = {use_pred_boxes: False, output_recall_scores: True}   (:506-646): IoU3D of the
            1000 proposals of every object with its ground-truth cube, the seven score functions, accumulate_scores, the 26
            combinations, the offset statistics, the chosen cube
  pseudo    training mode, pseudo_gt = 'pseudo'  (:462-490): the Instances written as pseudo ground truth
  learn     training mode, pseudo_gt = 'learn'   (:456-460): every proposal with its IoU2D score
  ap_gt     eval mode on GT boxes without output_recall_scores (:647-660)

and records the inputs (boxes, 3D ground truth, masks, depth / ground maps, the cubes `predict_cubes` sampled, the plane the
RANSAC stand-in returned, the numpy seed of the random score) with the outputs.

What is the reference's own code on this path: _forward_cube itself, accumulate_scores, Cubes / get_cuboid_verts_faces /
get_bube_corners / cubes_to_box, score_iou, score_dimensions, score_point_cloud, mat2euler, proposals.propose.  Third-party
arithmetic is stood in by this repo's restatements -- detectron2 Boxes / pairwise_iou, pyransac3d (a fixed plane), pytorch3d
box3d_overlap (oracle/iou3d.py, pinned by the reference's own 0.9944 vector), cv2 (minAreaRect: oracle/rect.py; convexHull +
fillPoly: oracle/geometry.segment_counts): parity unpinned at those boundaries.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_boxer.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _refimport  # noqa: E402

ns = _refimport.load()
d2 = importlib.import_module("3dod_amd.d2lite")
syn = importlib.import_module("3dod_amd.synthetic")
from oracle import geometry as og, rect as orect, iou3d as oiou  # noqa: E402

import cubercnn.modeling.roi_heads.roi_heads as ref_rh   # noqa: E402  (reference)

class Boxes(_refimport.Boxes):
    """+ the two detectron2 Boxes methods this path uses besides indexing / area [third-party, restated]"""
    @classmethod
    def cat(cls, boxes_list):
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

    def get_centers(self):
        return (self.tensor[:, :2] + self.tensor[:, 2:]) / 2

    def __getitem__(self, i):
        return Boxes(self.tensor[i].view(1, -1)) if isinstance(i, int) else Boxes(self.tensor[i])

    def __iter__(self):
        yield from self.tensor


_refimport.Boxes = Boxes
ns.conversions.Boxes = Boxes
ref_rh.Instances = d2.Instances
ref_rh.Boxes = Boxes
torch.set_num_threads(1)
PLANE = [0.03, -0.998, 0.05, 1.4]


class _FixedPlane:
    def fit(self, pts, thresh=0.05, maxIteration=1000):
        return list(PLANE), np.arange(3)


ref_rh.pyrsc = types.SimpleNamespace(Plane=_FixedPlane)


def _iou_3d(gt_cube, proposal_cubes):
    """ProposalNetwork/utils/utils.py:194-210 with pytorch3d.box3d_overlap -> oracle/iou3d.py"""
    g = gt_cube.get_all_corners()[0].numpy().astype(np.float64)
    p = proposal_cubes.get_all_corners()[0].numpy().astype(np.float64)
    return torch.as_tensor(oiou.box3d_overlap(g, p)[1][0], dtype=torch.float32)


def _score_corners(mask, bube_corners):
    """scorefunction.py:58-85 with cv2.findContours / minAreaRect / boxPoints -> oracle/rect.py"""
    c2 = bube_corners.squeeze(0).numpy()
    r = orect.rect_from_mask(mask.numpy().astype(bool))
    rect = og.fallback_rect(c2) if r is None else np.asarray(r, np.float32)
    return torch.as_tensor(og.score_corners_from_rect(rect, c2)[0])


def _seg(mask, bube_corners, mod):
    """scorefunction.py:88-126 with cv2.convexHull + fillPoly -> oracle/geometry.segment_counts"""
    m = mask.numpy().astype(np.uint8)
    cnt = og.segment_counts(bube_corners.squeeze(0).numpy(), m, 4)
    inter = cnt[:, 1].astype(np.float32)
    union = (cnt[:, 0] + (m[::4, ::4] != 0).sum() - cnt[:, 1]).astype(np.float32)
    val = inter ** 5 / np.maximum(union, 1) if mod else inter / np.maximum(union, 1)
    return torch.as_tensor(np.where(inter > 0, val, 0).astype(np.float32))


# Cubes.get_cubes builds pytorch3d meshes for the plots of tools/eval_boxes.py: the corners stand in for the mesh
ref_rh.Cubes.get_cubes = lambda self: [self.get_all_corners()[0, 0]]
ref_rh.iou_3d = _iou_3d
ref_rh.score_corners = _score_corners
ref_rh.score_segmentation = lambda m, c: _seg(m, c, False)
ref_rh.score_mod_segmentation = lambda m, c: _seg(m, c, True)


def make_case(seed, n_obj=5, P=1000, size=256):
    g = torch.Generator().manual_seed(seed)
    b = syn.make_batch(1, seed, size=size, min_obj=n_obj, max_obj=n_obj)[0]
    inst = b["instances"]
    K = torch.tensor(b["K"])
    v = torch.arange(size, dtype=torch.float32).view(-1, 1).expand(size, size)
    f = float(K[0, 0])
    depth = torch.where(v > size / 2 + 8, 1.5 * f / (v - size / 2).clamp(min=1.0), torch.full_like(v, 8.0)).clamp(max=8.0)
    depth = depth + torch.randn(size, size, generator=g) * 0.01
    ground = (v > size / 2 + 40)
    masks = torch.zeros(n_obj, 1, size, size, dtype=torch.bool)
    yy, xx = torch.arange(size)[:, None], torch.arange(size)[None, :]
    for j, bb in enumerate(inst.gt_boxes.tensor.tolist()):
        cx, cy, rx, ry = (bb[0] + bb[2]) / 2, (bb[1] + bb[3]) / 2, (bb[2] - bb[0]) / 2 + 0.5, (bb[3] - bb[1]) / 2 + 0.5
        if j != 1:                                   # object 1: an empty mask (the no-contour fallback of score_corners)
            masks[j, 0] = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1
    priors = torch.rand(1, 50, 2, 3, generator=g) * 0.8 + 0.3
    priors[:, :, 1] *= 0.3
    return b, inst, K, depth, ground, masks, priors, P, size


def run(seed, mode):
    b, inst, K, depth, ground, masks, priors, P, size = make_case(seed)
    self = types.SimpleNamespace(training=mode in ("pseudo", "learn"), dims_priors_enabled=True, priors_dims_per_cat=priors,
                                 number_of_proposals=P)
    C = ref_rh.ROIHeads_Boxer
    self.accumulate_scores = types.MethodType(C.accumulate_scores, self)
    rec_cubes = {}
    orig_predict = types.MethodType(C.predict_cubes, self)

    def predict(*a, **k):
        out = orig_predict(*a, **k)
        rec_cubes["cubes"] = out[0].tensor.clone()
        rec_cubes["stats_image"], rec_cubes["stats_ranges"] = out[2], out[3]
        return out
    self.predict_cubes = predict
    images = torch.zeros(1, 3, size, size)
    images_raw = d2.ImageList(torch.zeros(1, 3, size, size), [(size, size)])
    depth_maps = d2.ImageList(depth[None], [(size, size)])
    ground_maps = d2.ImageList(ground[None], [(size, size)])
    gi = d2.Instances((size, size))
    gi.gt_boxes = Boxes(inst.gt_boxes.tensor)
    gi.gt_classes, gi.gt_boxes3D, gi.gt_poses = inst.gt_classes, inst.gt_boxes3D, inst.gt_poses
    ex = {"use_pred_boxes": False}
    if mode == "mabo":
        ex["output_recall_scores"] = True
    if mode in ("pseudo", "learn"):
        ex["pseudo_gt"] = mode
    torch.manual_seed(seed)
    np.random.seed(seed)
    out = C._forward_cube(self, images, images_raw, None, [masks], depth_maps, ground_maps, None, [gi], [K], [(size, size)], [1.0],
                          ex, "propose")
    rec = dict(seed=np.array(seed), mode=np.array(mode), K=K.numpy(), depth=depth.numpy(), ground=ground.numpy(), masks=masks[:, 0].numpy(),
               priors=priors.numpy(), gt_boxes=inst.gt_boxes.tensor.numpy(), gt_classes=inst.gt_classes.numpy(),
               gt_boxes3D=inst.gt_boxes3D.numpy(), gt_poses=inst.gt_poses.numpy(), plane=np.array(PLANE, np.float32),
               cubes=rec_cubes["cubes"].numpy(), stats_image=np.asarray(rec_cubes["stats_image"], dtype=np.float64),
               stats_ranges=np.asarray(rec_cubes["stats_ranges"], dtype=np.float64))
    if mode == "mabo":
        (p_info, s_iou, s_seg, s_dim, s_comb, s_rand, s_pc, empty, stats_image, stats_off, s_segm, s_cor, comb) = out
        rec.update(score_IoU2D=s_iou, score_seg=s_seg, score_dim=s_dim, score_combined=s_comb, score_random=s_rand,
                   score_point_c=s_pc, stat_empty_boxes=np.array(empty), stats_off=stats_off, score_seg_mod=s_segm,
                   score_corner=s_cor, combinations=comb, out_cubes=p_info.pred_cubes.tensor.numpy(),
                   out_scores=p_info.pred_cubes.scores.numpy(), out_pred_boxes=p_info.pred_boxes.tensor.numpy())
    elif mode == "learn":
        rec.update(out_cubes=out.tensor.numpy(), out_scores=out.scores.numpy())
    else:
        r = out[0]
        for f in ("scores", "pred_classes", "pred_bbox3D", "pred_center_cam", "pred_dimensions", "pred_pose", "pred_center_2D"):
            rec["out_" + f] = r.get(f).detach().numpy()
        rec["out_pred_boxes"] = r.pred_boxes.tensor.numpy()
    rec["notes"] = ("reference ROIHeads_Boxer._forward_cube (roi_heads.py:304-660), mode %s; stand-ins for detectron2 Boxes / "
                    "pairwise_iou, pyransac3d (fixed plane), pytorch3d box3d_overlap (oracle/iou3d.py), cv2 (oracle/rect.py, "
                    "oracle/geometry.segment_counts): parity unpinned at those boundaries" % mode)
    return rec


if __name__ == "__main__":
    for mode, seed in (("mabo", 31), ("pseudo", 32), ("learn", 33), ("ap_gt", 34)):
        r = run(seed, mode)
        np.savez_compressed(os.path.join(HERE, "boxer_%s.npz" % mode), **r)
        print(mode, {k: np.asarray(v).shape for k, v in r.items() if k.startswith(("score_", "out_", "comb", "stats_"))})
