"""CPU restatements for the weakly supervised 3D head -- TEST INFRASTRUCTURE ONLY (tests/, smoke): the product path
uses cr_box_median and cr_ransac_plane and has no CPU fallback.

box_median: the per-box torch.median of ROIHeads3DScore.pseudo_gt_z_box_loss
(cubercnn/modeling/roi_heads/roi_heads.py:1216-1218).  Plane: Plane.fit_parallel
(ProposalNetwork/utils/plane.py:79-134) through oracle/geometry.ransac_plane, with the triples given or drawn like the
reference (random.sample).  Pinned by tests/golden/weakhead_*.npz (outputs of the reference's _forward_cube)."""
import random

import numpy as np
import torch

from . import geometry as og


def box_median(depth, boxes, img):
    """lower median of depth[img, y1:y2, x1:x2]; NaN for an empty window (torch.median raises there)."""
    out = []
    for (x1, y1, x2, y2), i in zip(boxes.tolist(), img.tolist()):
        w = depth[i, max(y1, 0):max(y2, 0), max(x1, 0):max(x2, 0)]
        out.append(torch.median(w) if w.numel() else depth.new_full((), float("nan")))
    return torch.stack(out) if out else depth.new_zeros(0)


class Plane:
    def __init__(self):
        self.inliers, self.equation = [], []

    def fit_parallel(self, pts, thresh=0.05, minPoints=100, maxIteration=1000, id_samples=None, generator=None,
                     need_inliers=True):
        n = pts.shape[0]
        if id_samples is None:
            id_samples = [random.sample(range(0, n), 3) for _ in range(maxIteration)]
        neg_eq, _, _, _ = og.ransac_plane(pts.detach().cpu().numpy(), np.asarray(id_samples), thresh)
        neg_eq = torch.as_tensor(neg_eq, dtype=torch.float32, device=pts.device)
        eq = -neg_eq
        dist = (eq[0] * pts[:, 0] + eq[1] * pts[:, 1] + eq[2] * pts[:, 2] + eq[3]) / torch.sqrt(eq[0] ** 2 + eq[1] ** 2 + eq[2] ** 2)
        self.inliers, self.equation = torch.where(torch.abs(dist) <= thresh)[0], eq
        return neg_eq, self.inliers
