"""ORACLE (test infrastructure, NOT product code) -- plain PyTorch float32 CPU
restatements of the conv / BatchNorm / pooling / ROIAlign / NMS / SGD ops that the
HIP kernels implement.  NCHW like the reference; tests convert layouts.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.

ROIAlign, NMS, FPN level assignment and box coding live in torchvision /
detectron2, which are absent from the reference tree and not installed: they
are restated here from their published definitions and marked
"parity unpinned" (SURVEY.md 8c); everything else is torch's own ATen op.
"""
import math

import torch
import torch.nn.functional as F


def conv_bn_act(x, w, gamma, beta, stride, pad, relu=True, residual=None, eps=1e-5):
    """dla.py:40-68,156-174: conv (no bias) -> BatchNorm2d(train) -> (+residual) -> ReLU."""
    y = F.conv2d(x, w, None, stride, pad)
    y = F.batch_norm(y, None, None, gamma, beta, True, 0.1, eps)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def conv_bias_act(x, w, b, stride, pad, relu=False):
    y = F.conv2d(x, w, b, stride, pad)
    return F.relu(y) if relu else y


def upsample2x_add(lat, top):
    """detectron2 FPN top-down step [3rd-party, restated]."""
    return lat + F.interpolate(top, scale_factor=2.0, mode="nearest")


def assign_levels(boxes, min_level=2, max_level=6, canonical_size=224, canonical_level=4):
    """detectron2 assign_boxes_to_levels [3rd-party, restated]."""
    sizes = torch.sqrt((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))
    lv = torch.floor(canonical_level + torch.log2(sizes / canonical_size + 1e-8))
    return torch.clamp(lv, min=min_level, max=max_level).to(torch.int64) - min_level


def _bilinear(feat, y, x):
    """feat (C,H,W); y,x scalars (python floats).  torchvision roi_align bilinear_interpolate."""
    C, H, W = feat.shape
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return torch.zeros(C, dtype=feat.dtype)
    y = max(y, 0.0)
    x = max(x, 0.0)
    yl, xl = int(y), int(x)
    if yl >= H - 1:
        yh = yl = H - 1
        y = float(yl)
    else:
        yh = yl + 1
    if xl >= W - 1:
        xh = xl = W - 1
        x = float(xl)
    else:
        xh = xl + 1
    ly, lx = y - yl, x - xl
    hy, hx = 1.0 - ly, 1.0 - lx
    return hy * hx * feat[:, yl, xl] + hy * lx * feat[:, yl, xh] + ly * hx * feat[:, yh, xl] + ly * lx * feat[:, yh, xh]


def roi_align(feats, rois, scales, out_size):
    """torchvision.ops.roi_align(aligned=True, sampling_ratio=0) over an FPN pyramid with detectron2's level
    assignment [3rd-party, restated; differentiable through torch ops].  feats: list of (N,C,H,W);
    rois (R,5) [batch,x1,y1,x2,y2] -> (R,C,out,out)."""
    R = rois.shape[0]
    C = feats[0].shape[1]
    min_level = int(round(-math.log2(scales[0])))
    lv = assign_levels(rois[:, 1:], min_level, min_level + len(feats) - 1)
    out = []
    for r in range(R):
        l = int(lv[r])
        f = feats[l][int(rois[r, 0])]
        s = scales[l]
        x1, y1 = float(rois[r, 1]) * s - 0.5, float(rois[r, 2]) * s - 0.5
        rw, rh = float(rois[r, 3] - rois[r, 1]) * s, float(rois[r, 4] - rois[r, 2]) * s
        bw, bh = rw / out_size, rh / out_size
        gh, gw = int(math.ceil(rh / out_size)), int(math.ceil(rw / out_size))
        cnt = max(gh * gw, 1)
        bins = []
        for ph in range(out_size):
            for pw in range(out_size):
                acc = torch.zeros(C, dtype=f.dtype)
                for iy in range(gh):
                    yy = y1 + ph * bh + (iy + 0.5) * bh / gh
                    for ix in range(gw):
                        xx = x1 + pw * bw + (ix + 0.5) * bw / gw
                        acc = acc + _bilinear(f, yy, xx)
                bins.append(acc / cnt)
        out.append(torch.stack(bins, 1).view(C, out_size, out_size))
    return torch.stack(out) if out else torch.zeros((0, C, out_size, out_size))


def nms(boxes, scores, thresh):
    """torchvision.ops.nms [3rd-party, restated]: greedy, descending score, suppress IoU > thresh.
    Returns kept indices sorted by descending score."""
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    n = b.shape[0]
    dead = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if dead[i]:
            continue
        keep.append(i)
        if i + 1 < n:
            lt = torch.max(b[i, :2], b[i + 1:, :2])
            rb = torch.min(b[i, 2:], b[i + 1:, 2:])
            wh = (rb - lt).clamp(min=0)
            inter = wh[:, 0] * wh[:, 1]
            iou = inter / (area[i] + area[i + 1:] - inter)
            dead[i + 1:] |= iou > thresh
    return order[torch.tensor(keep, dtype=torch.int64)]


def sgd_step(p, g, m, lr, momentum, wd):
    """torch.optim.SGD update (dampening 0, no nesterov), cubercnn/solver/build.py:50-56."""
    gg = g + wd * p
    m_new = momentum * m + gg
    return p - lr * m_new, m_new
