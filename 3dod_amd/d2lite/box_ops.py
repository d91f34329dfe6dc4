"""Box coding, matcher, anchor generator (detectron2.modeling.box_regression / matcher /
anchor_generator) [third-party behaviour restated]."""
import math
from typing import List

import torch

from .structures import Boxes

_DEFAULT_SCALE_CLAMP = math.log(1000.0 / 16)


class Box2BoxTransform:
    def __init__(self, weights, scale_clamp=_DEFAULT_SCALE_CLAMP):
        self.weights = weights
        self.scale_clamp = scale_clamp

    def get_deltas(self, src_boxes, target_boxes):
        sw = src_boxes[:, 2] - src_boxes[:, 0]
        sh = src_boxes[:, 3] - src_boxes[:, 1]
        sx = src_boxes[:, 0] + 0.5 * sw
        sy = src_boxes[:, 1] + 0.5 * sh
        tw = target_boxes[:, 2] - target_boxes[:, 0]
        th = target_boxes[:, 3] - target_boxes[:, 1]
        tx = target_boxes[:, 0] + 0.5 * tw
        ty = target_boxes[:, 1] + 0.5 * th
        wx, wy, ww, wh = self.weights
        dx = wx * (tx - sx) / sw
        dy = wy * (ty - sy) / sh
        dw = ww * torch.log(tw / sw)
        dh = wh * torch.log(th / sh)
        return torch.stack((dx, dy, dw, dh), dim=1)

    def apply_deltas(self, deltas, boxes):
        deltas = deltas.float()
        boxes = boxes.to(deltas.dtype)
        widths = boxes[:, 2] - boxes[:, 0]
        heights = boxes[:, 3] - boxes[:, 1]
        ctr_x = boxes[:, 0] + 0.5 * widths
        ctr_y = boxes[:, 1] + 0.5 * heights
        wx, wy, ww, wh = self.weights
        dx = deltas[:, 0::4] / wx
        dy = deltas[:, 1::4] / wy
        dw = deltas[:, 2::4] / ww
        dh = deltas[:, 3::4] / wh
        dw = torch.clamp(dw, max=self.scale_clamp)
        dh = torch.clamp(dh, max=self.scale_clamp)
        pred_ctr_x = dx * widths[:, None] + ctr_x[:, None]
        pred_ctr_y = dy * heights[:, None] + ctr_y[:, None]
        pred_w = torch.exp(dw) * widths[:, None]
        pred_h = torch.exp(dh) * heights[:, None]
        x1 = pred_ctr_x - 0.5 * pred_w
        y1 = pred_ctr_y - 0.5 * pred_h
        x2 = pred_ctr_x + 0.5 * pred_w
        y2 = pred_ctr_y + 0.5 * pred_h
        pred_boxes = torch.stack((x1, y1, x2, y2), dim=-1)
        return pred_boxes.reshape(deltas.shape)


class Matcher:
    def __init__(self, thresholds: List[float], labels: List[int], allow_low_quality_matches: bool = False):
        thresholds = thresholds[:]
        assert thresholds[0] > 0
        thresholds.insert(0, -float("inf"))
        thresholds.append(float("inf"))
        assert all(low <= high for (low, high) in zip(thresholds[:-1], thresholds[1:]))
        assert all(l in [-1, 0, 1] for l in labels)
        assert len(labels) == len(thresholds) - 1
        self.thresholds = thresholds
        self.labels = labels
        self.allow_low_quality_matches = allow_low_quality_matches

    def __call__(self, match_quality_matrix):
        assert match_quality_matrix.dim() == 2
        if match_quality_matrix.numel() == 0:
            default_matches = match_quality_matrix.new_full((match_quality_matrix.size(1),), 0, dtype=torch.int64)
            default_match_labels = match_quality_matrix.new_full((match_quality_matrix.size(1),), self.labels[0],
                                                                 dtype=torch.int8)
            return default_matches, default_match_labels
        assert torch.all(match_quality_matrix >= 0)
        matched_vals, matches = match_quality_matrix.max(dim=0)
        match_labels = matches.new_full(matches.size(), 1, dtype=torch.int8)
        for (l, low, high) in zip(self.labels, self.thresholds[:-1], self.thresholds[1:]):
            low_high = (matched_vals >= low) & (matched_vals < high)
            match_labels[low_high] = l
        if self.allow_low_quality_matches:
            highest_quality_foreach_gt, _ = match_quality_matrix.max(dim=1)
            pred_inds = (match_quality_matrix == highest_quality_foreach_gt[:, None]).nonzero()[:, 1]
            match_labels[pred_inds] = 1
        return matches, match_labels


class DefaultAnchorGenerator(torch.nn.Module):
    box_dim = 4

    def __init__(self, sizes, aspect_ratios, strides, offset=0.0):
        super().__init__()
        self.strides = strides
        n = len(strides)
        sizes = list(sizes) * n if len(sizes) == 1 else list(sizes)
        aspect_ratios = list(aspect_ratios) * n if len(aspect_ratios) == 1 else list(aspect_ratios)
        assert len(sizes) == n and len(aspect_ratios) == n
        self.cell_anchors = [self.generate_cell_anchors(s, a).float() for s, a in zip(sizes, aspect_ratios)]
        self.offset = offset
        self._cache = {}

    @classmethod
    def from_config(cls, cfg, input_shape):
        return cls(cfg.MODEL.ANCHOR_GENERATOR.SIZES, cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS,
                   [x.stride for x in input_shape], cfg.MODEL.ANCHOR_GENERATOR.OFFSET)

    @property
    def num_anchors(self):
        return [len(c) for c in self.cell_anchors]

    @staticmethod
    def generate_cell_anchors(sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2)):
        anchors = []
        for size in sizes:
            area = size ** 2.0
            for aspect_ratio in aspect_ratios:
                w = math.sqrt(area / aspect_ratio)
                h = aspect_ratio * w
                anchors.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
        return torch.tensor(anchors)

    def grid_anchors(self, grid_sizes, device):
        out = []
        for size, stride, base in zip(grid_sizes, self.strides, self.cell_anchors):
            key = (tuple(size), stride, str(device))
            a = self._cache.get(key)
            if a is None:
                gh, gw = size
                sx = torch.arange(self.offset * stride, gw * stride, step=stride, dtype=torch.float32, device=device)
                sy = torch.arange(self.offset * stride, gh * stride, step=stride, dtype=torch.float32, device=device)
                shift_y, shift_x = torch.meshgrid(sy, sx, indexing="ij")
                shift_x = shift_x.reshape(-1)
                shift_y = shift_y.reshape(-1)
                shifts = torch.stack((shift_x, shift_y, shift_x, shift_y), dim=1)
                a = (shifts.view(-1, 1, 4) + base.to(device).view(1, -1, 4)).reshape(-1, 4)
                self._cache[key] = a
            out.append(a)
        return out

    def forward(self, grid_sizes, device):
        """grid_sizes: list of (H,W) per level -> list[Boxes]"""
        return [Boxes(a) for a in self.grid_anchors(grid_sizes, device)]


def subsample_labels_d2(labels, num_samples, positive_fraction, bg_label):
    """detectron2.modeling.sampling.subsample_labels (uniform randperm variant)."""
    positive = ((labels != -1) & (labels != bg_label)).nonzero().squeeze(1)
    negative = (labels == bg_label).nonzero().squeeze(1)
    num_pos = min(positive.numel(), int(num_samples * positive_fraction))
    num_neg = min(negative.numel(), num_samples - num_pos)
    perm1 = torch.randperm(positive.numel(), device=positive.device)[:num_pos]
    perm2 = torch.randperm(negative.numel(), device=negative.device)[:num_neg]
    return positive[perm1], negative[perm2]
