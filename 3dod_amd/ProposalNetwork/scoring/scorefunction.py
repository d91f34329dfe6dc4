"""Scoring functions of ProposalNetwork/scoring/scorefunction.py:47-85,144-160 on the fused kernel.
score_all is the batched form used by the model; the per-object functions keep the reference's signatures."""
import torch

from ... import geometry as geo
from ...d2lite import Boxes
from ..utils.utils import iou_2d


def score_all(cubes, K, im_shape, gt_boxes, prior_mean, prior_std, rect_pts=None, want=("iou", "dim", "corner", "combined")):
    """project + IoU + size prior + corner chamfer + product + argmax for N objects x P proposals in ONE launch
    (roi_heads.py:492-505).  gt_boxes: Boxes or (N,4) tensor.  rect_pts (N,4,2) = cv2.boxPoints(minAreaRect(mask))
    per object, or None for the reference's no-contour fallback."""
    ref = gt_boxes.tensor if isinstance(gt_boxes, Boxes) else gt_boxes
    return geo.cubes_project_score(cubes.tensor.contiguous(), K, im_shape, ref.contiguous(), prior_mean.contiguous(),
                                   prior_std.contiguous(), rect_pts, want=want)


def score_iou(gt_box, proposal_box):
    """scorefunction.py:47-49."""
    return iou_2d(gt_box, proposal_box)


def score_dimensions(category, dimensions, gt_boxes, pred_boxes):
    """scorefunction.py:144-160 (torch form for a single object; the fused kernel computes the same)."""
    prior_mean, prior_std = category
    dimensions_scores = torch.exp(-1 / 2 * ((dimensions - prior_mean) / prior_std) ** 2)
    scores = dimensions_scores.mean(1)
    gt_ratio = (gt_boxes.tensor[0, 2] - gt_boxes.tensor[0, 0]) / (gt_boxes.tensor[0, 3] - gt_boxes.tensor[0, 1])
    pred_ratios = (pred_boxes.tensor[:, 2] - pred_boxes.tensor[:, 0]) / (pred_boxes.tensor[:, 3] - pred_boxes.tensor[:, 1])
    differences = torch.abs(gt_ratio - pred_ratios)
    return (1 - differences / torch.max(differences)) * scores


def score_point_cloud(point_cloud, cubes, K=None, segmentation_mask=None):
    """scorefunction.py:9-43 (MABO diagnostics only, roi_heads.py:535): number of cloud points inside a box derived from
    the cubes' corners.  Mirrors the reference expression exactly -- it reads the bounds off `verts[:, i].min(1)` /
    `.max(1)` for i = 0,1,2, i.e. the smallest / largest coordinate of corners 0, 1 and 2 of each cube, not the per-axis
    extent over the eight corners -- so that MABO numbers stay comparable.  point_cloud (Q,3), cubes: Cubes with one
    object (1,P,15) -> (P,) int64."""
    verts = cubes.get_all_corners().squeeze(0)                                # (P,8,3)
    lo = [verts[:, i].min(1)[0] for i in range(3)]
    hi = [verts[:, i].max(1)[0] for i in range(3)]
    pc = point_cloud
    inside = (pc[:, 0].view(-1, 1) > lo[0]) & (pc[:, 0].view(-1, 1) < hi[0]) & \
             (pc[:, 1].view(-1, 1) > lo[1]) & (pc[:, 1].view(-1, 1) < hi[1]) & \
             (pc[:, 2].view(-1, 1) > lo[2]) & (pc[:, 2].view(-1, 1) < hi[2])
    return inside.sum(0)


def _segment_terms(segmentation_mask, bube_corners, counts_fn=None):
    """intersection / union of every proposal's rasterised hull with the object mask on the [::4, ::4] grid"""
    corners = bube_corners.to(device=segmentation_mask.device).squeeze(0)
    counts = (counts_fn or geo.segment_counts)(corners, segmentation_mask, 4)
    inter = counts[:, 1].to(torch.float32)
    union = (counts[:, 0] + (segmentation_mask[::4, ::4] != 0).sum() - counts[:, 1]).to(torch.float32)
    return inter, union


def score_segmentation(segmentation_mask, bube_corners, counts_fn=None):
    """scorefunction.py:88-105 with mask_iou (utils.py:230-239), MABO only: IoU of the proposal's filled convex hull and
    the object mask, both sampled every 4th pixel; 0 where they do not intersect.  (1,P,8,2) corners -> (P,)"""
    inter, union = _segment_terms(segmentation_mask, bube_corners, counts_fn)
    return torch.where(inter > 0, inter / union.clamp(min=1), torch.zeros_like(inter))


def score_mod_segmentation(segmentation_mask, bube_corners, counts_fn=None):
    """scorefunction.py:107-124 with mod_mask_iou (utils.py:241-250): intersection^5 / union (not a standard IoU)"""
    inter, union = _segment_terms(segmentation_mask, bube_corners, counts_fn)
    return torch.where(inter > 0, inter ** 5 / union.clamp(min=1), torch.zeros_like(inter))
