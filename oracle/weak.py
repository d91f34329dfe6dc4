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


def hull8(points):
    """jarvis_march (ProposalNetwork/utils/utils.py:424-470) restated in plain Python per RoI, float32 arithmetic on
    tensors like the reference: -> order (n,8) int64, count (n) int32, bump (n,8) float32"""
    P = points.detach().float().cpu()
    n = P.shape[0]
    order = torch.zeros((n, 8), dtype=torch.int64)
    count = torch.zeros((n,), dtype=torch.int32)
    bump = torch.zeros((n, 8), dtype=torch.float32)
    for r in range(n):
        pts = P[r].clone()
        dups = [i for i in range(7) if any(bool(torch.all(pts[i] == pts[j])) for j in range(i + 1, 8))]
        for k, d in enumerate(dups):
            bump[r, d] = k + 1
        pts = pts + bump[r][:, None]
        minx = pts[:, 0].min()
        cand = (pts[:, 0] == minx).nonzero(as_tuple=True)[0]
        start = int(cand[torch.argmax(pts[cand][:, 1])]) if len(cand) > 1 else int(cand[0])
        res, l = [start], start
        for _ in range(8):
            q = (l + 1) % 8
            for i in range(8):
                if i == l:
                    continue
                d = (pts[i, 0] - pts[l, 0]) * (pts[q, 1] - pts[l, 1]) - (pts[i, 1] - pts[l, 1]) * (pts[q, 0] - pts[l, 0])
                di = (pts[i, 0] - pts[l, 0]) ** 2 + (pts[i, 1] - pts[l, 1]) ** 2
                dq = (pts[q, 0] - pts[l, 0]) ** 2 + (pts[q, 1] - pts[l, 1]) ** 2
                if d > 0 or (d == 0 and di > dq):
                    q = i
            l = q
            if l == start or len(res) >= 8:
                break
            res.append(q)
        res = res[::-1]
        count[r] = len(res)
        order[r, :len(res)] = torch.tensor(res)
    dev = points.device
    return order.to(dev), count.to(dev), bump.to(dev)


def polygon_focal(hull, count, masks, mask_idx):
    """fill_polygon (utils.py:472-502) + torchvision sigmoid_focal_loss(inputs = mask, targets = polygon; alpha 0.25,
    gamma 2) [third-party, restated], mean over pixels; differentiable through autograd"""
    import torch.nn.functional as F
    n = hull.shape[0]
    H, W = masks.shape[1:]
    Y, X = torch.meshgrid(torch.arange(H, device=hull.device), torch.arange(W, device=hull.device), indexing='ij')
    X, Y = X.float(), Y.float()
    out = []
    for r in range(n):
        k = int(count[r])
        m = torch.ones(H, W, device=hull.device)
        for e in range(k):
            v1, v2 = hull[r, e], hull[r, (e + 1) % k]
            ed = v2 - v1
            raw = (X - v1[0]) * ed[1] - (Y - v1[1]) * ed[0]
            m = m * torch.min(torch.max(raw, torch.zeros_like(raw)), torch.ones_like(raw))
        x = masks[int(mask_idx[r])].float()
        p = torch.sigmoid(x)
        ce = F.binary_cross_entropy_with_logits(x, m, reduction="none")
        p_t = p * m + (1 - p) * (1 - m)
        loss = ce * ((1 - p_t) ** 2)
        loss = (0.25 * m + 0.75 * (1 - m)) * loss
        out.append(loss.mean())
    return torch.stack(out)
