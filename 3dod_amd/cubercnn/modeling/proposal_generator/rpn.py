"""RPN with ignore regions and the "IoUness" objectness loss.

RPNWithIgnore restates cubercnn/modeling/proposal_generator/rpn.py:19-354 of the reference; the base RPN
(head, anchors, decode, per-level top-k, NMS) restates detectron2's RPN / StandardRPNHead [third-party].
The head convolutions run on cr_conv2d_* (shared weights over the 5 levels; the objectness and delta 1x1
predictors are evaluated as ONE 16-channel conv), NMS on cr_nms_grouped."""
from typing import Dict, List, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ....d2lite import (PROPOSAL_GENERATOR_REGISTRY, RPN_HEAD_REGISTRY, Boxes, Instances, ShapeSpec, Box2BoxTransform,
                        Matcher, DefaultAnchorGenerator, cat, pairwise_iou, pairwise_ioa, get_event_storage)
from .... import hipops as ops
from ..backbone.fpn import to_channels_last


@RPN_HEAD_REGISTRY.register()
class StandardRPNHead(nn.Module):
    def __init__(self, in_channels, num_anchors, box_dim=4):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)
        self.objectness_logits = nn.Conv2d(in_channels, num_anchors, kernel_size=1, stride=1)
        self.anchor_deltas = nn.Conv2d(in_channels, num_anchors * box_dim, kernel_size=1, stride=1)
        for layer in (self.conv, self.objectness_logits, self.anchor_deltas):
            nn.init.normal_(layer.weight, std=0.01)
            nn.init.constant_(layer.bias, 0)
        self.num_anchors, self.box_dim = num_anchors, box_dim
        to_channels_last(self)

    def forward_raw(self, features: List[torch.Tensor]):
        """per level y (N,H,W,16) f32 = [A objectness logits | 4A anchor deltas | padding]: both predictors as ONE conv"""
        A, D = self.num_anchors, self.box_dim
        n_out = A + A * D
        pad = (-n_out) % 16
        w = ops.cat_rows((self.objectness_logits.weight, self.anchor_deltas.weight), n_out + pad)
        b = ops.cat_rows((self.objectness_logits.bias, self.anchor_deltas.bias), n_out + pad)
        # the 3x3 convolution is ONE set of weights applied to every level: one grouped launch per direction, the weight
        # gradient of all levels summed in one pass
        n = len(features)
        ws, bs = [self.conv.weight] * n, [self.conv.bias] * n
        if n > 1 and self.stack_levels and ops.group_supported(list(features), [ops.as_krsc(x) for x in ws]):
            # the levels' outputs as row blocks of ONE map: the two 1x1 predictors then run once over all levels (a 1x1
            # convolution does not care which level a pixel belongs to) -- one forward, one backward-data and one weight
            # gradient launch instead of five each, and no fan-in adds for the shared predictor parameters.  The result is
            # a one-element list holding Y (1, 1, sum_l N H_l W_l, 16); `level_views` cuts the per-level maps out of it
            t = ops.conv_bias_act_group(list(features), ws, bs, pad=1, relu=True, stacked=True)
            return [ops.conv_bias_act(t, w, b, 1, 0, relu=False, out_f32=True)]
        ts = ops.conv_bias_act_group(list(features), ws, bs, pad=1, relu=True)
        return [ops.conv_bias_act(t, w, b, 1, 0, relu=False, out_f32=True) for t in ts]

    stack_levels = True

    @staticmethod
    def level_views(ys, features):
        """the per-level maps (N,H_l,W_l,16) of forward_raw's result, stacked or not (views, no copies)"""
        if len(ys) == len(features):
            return list(ys)
        flat, out, off = ys[0].view(-1, ys[0].shape[-1]), [], 0
        for f in features:
            m = f.shape[0] * f.shape[1] * f.shape[2]
            out.append(flat[off:off + m].view(f.shape[0], f.shape[1], f.shape[2], -1))
            off += m
        return out

    def forward(self, features: List[torch.Tensor]):
        A, D = self.num_anchors, self.box_dim
        n_out = A + A * D
        pred_objectness_logits, pred_anchor_deltas = [], []
        for y in self.level_views(self.forward_raw(features), features):
            N = y.shape[0]
            pred_objectness_logits.append(y[..., :A].reshape(N, -1))                    # (N, H*W*A)
            pred_anchor_deltas.append(y[..., A:n_out].reshape(N, -1, D))                # (N, H*W*A, 4)
        return pred_objectness_logits, pred_anchor_deltas


_CONST = {}


def find_top_rpn_proposals(proposals, pred_objectness_logits, image_sizes, nms_thresh, pre_nms_topk, post_nms_topk,
                           min_box_size, training, padded=False):
    """detectron2 find_top_rpn_proposals [third-party, restated]: per level top-k, clip, drop empty boxes,
    per-level NMS, then the post_nms_topk best per image."""
    num_images = len(image_sizes)
    device = proposals[0].device
    L = len(proposals)
    ks = [min(p.shape[1], pre_nms_topk) for p in proposals]
    maxn = max(ks)
    boxes_pad = proposals[0].new_zeros((num_images, L, maxn, 4))
    scores_pad = proposals[0].new_full((num_images, L, maxn), float("-inf"))
    for l, (props, logits, k) in enumerate(zip(proposals, pred_objectness_logits, ks)):
        topk_scores, topk_idx = ops.topk(logits.float(), k)        # csrc/topk.hip (ties -> lower index, like a stable sort)
        boxes_pad[:, l, :k] = torch.gather(props, 1, topk_idx[:, :, None].expand(-1, -1, 4))
        scores_pad[:, l, :k] = topk_scores
    ckey = (tuple(tuple(s) for s in image_sizes), tuple(ks), str(device))
    cached = _CONST.get(ckey)
    if cached is None:       # constant per (image sizes, level sizes): made once, never inside a captured region
        cached = (torch.tensor([[s[1], s[0], s[1], s[0]] for s in image_sizes], dtype=torch.float32, device=device),
                  torch.tensor(ks, dtype=torch.int32, device=device).repeat(num_images))
        _CONST[ckey] = cached
    hw, counts = cached
    finite = torch.isfinite(boxes_pad).all(dim=3) & torch.isfinite(scores_pad)
    boxes_pad = torch.where(finite[..., None], boxes_pad, torch.zeros((), device=device))
    boxes_pad = torch.minimum(boxes_pad.clamp(min=0), hw[:, None, None, :])              # Boxes.clip
    valid = finite & ((boxes_pad[..., 2] - boxes_pad[..., 0]) > min_box_size) & \
        ((boxes_pad[..., 3] - boxes_pad[..., 1]) > min_box_size)
    # invalid boxes become zero-area: they neither suppress nor survive
    nms_boxes = torch.where(valid[..., None], boxes_pad, torch.zeros((), device=device))
    keep = ops.nms_grouped(nms_boxes.view(num_images * L, maxn, 4), counts, nms_thresh).view(num_images, L, maxn)
    keep = keep & valid
    flat_scores = torch.where(keep, scores_pad, torch.full((), float("-inf"), device=device)).view(num_images, -1)
    flat_boxes = boxes_pad.view(num_images, -1, 4)
    k_post = min(post_nms_topk, flat_scores.shape[1])
    top_scores, top_idx = ops.topk(flat_scores, k_post)
    if padded:       # fixed-shape result for the sync-free training path: empty slots carry score -inf
        return torch.gather(flat_boxes, 1, top_idx[:, :, None].expand(-1, -1, 4)), top_scores
    n_keep = keep.view(num_images, -1).sum(1).clamp(max=k_post).tolist()                 # one host sync
    results = []
    for i in range(num_images):
        res = Instances(image_sizes[i])
        idx = top_idx[i, :n_keep[i]]
        res.proposal_boxes = Boxes(flat_boxes[i][idx])
        res.objectness_logits = top_scores[i, :n_keep[i]]
        results.append(res)
    return results


class RPN(nn.Module):
    """detectron2 RPN [third-party, restated]."""

    def __init__(self, *, in_features, head, anchor_generator, anchor_matcher, box2box_transform, batch_size_per_image,
                 positive_fraction, pre_nms_topk, post_nms_topk, nms_thresh=0.7, min_box_size=0.0,
                 anchor_boundary_thresh=-1.0, loss_weight=1.0, box_reg_loss_type="smooth_l1", smooth_l1_beta=0.0):
        super().__init__()
        self.in_features = in_features
        self.rpn_head = head
        self.anchor_generator = anchor_generator
        self.anchor_matcher = anchor_matcher
        self.box2box_transform = box2box_transform
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.pre_nms_topk = {True: pre_nms_topk[0], False: pre_nms_topk[1]}
        self.post_nms_topk = {True: post_nms_topk[0], False: post_nms_topk[1]}
        self.nms_thresh = nms_thresh
        self.min_box_size = float(min_box_size)
        self.anchor_boundary_thresh = anchor_boundary_thresh
        if isinstance(loss_weight, float):
            loss_weight = {"loss_rpn_cls": loss_weight, "loss_rpn_loc": loss_weight}
        self.loss_weight = loss_weight
        self.box_reg_loss_type = box_reg_loss_type
        self.smooth_l1_beta = smooth_l1_beta

    @classmethod
    def from_config(cls, cfg, input_shape: Dict[str, ShapeSpec]):
        in_features = cfg.MODEL.RPN.IN_FEATURES
        shapes = [input_shape[f] for f in in_features]
        anchor_generator = DefaultAnchorGenerator.from_config(cfg, shapes)
        num_anchors = anchor_generator.num_anchors
        assert len(set(num_anchors)) == 1
        in_channels = shapes[0].channels
        head = RPN_HEAD_REGISTRY.get(cfg.MODEL.RPN.HEAD_NAME)(in_channels, num_anchors[0], 4)
        return {
            "in_features": in_features,
            "min_box_size": cfg.MODEL.PROPOSAL_GENERATOR.MIN_SIZE,
            "nms_thresh": cfg.MODEL.RPN.NMS_THRESH,
            "batch_size_per_image": cfg.MODEL.RPN.BATCH_SIZE_PER_IMAGE,
            "positive_fraction": cfg.MODEL.RPN.POSITIVE_FRACTION,
            "loss_weight": {"loss_rpn_cls": cfg.MODEL.RPN.LOSS_WEIGHT,
                            "loss_rpn_loc": cfg.MODEL.RPN.BBOX_REG_LOSS_WEIGHT * cfg.MODEL.RPN.LOSS_WEIGHT},
            "anchor_boundary_thresh": cfg.MODEL.RPN.BOUNDARY_THRESH,
            "box2box_transform": Box2BoxTransform(weights=cfg.MODEL.RPN.BBOX_REG_WEIGHTS),
            "box_reg_loss_type": cfg.MODEL.RPN.BBOX_REG_LOSS_TYPE,
            "smooth_l1_beta": cfg.MODEL.RPN.SMOOTH_L1_BETA,
            "pre_nms_topk": (cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, cfg.MODEL.RPN.PRE_NMS_TOPK_TEST),
            "post_nms_topk": (cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN, cfg.MODEL.RPN.POST_NMS_TOPK_TEST),
            "anchor_generator": anchor_generator,
            "anchor_matcher": Matcher(cfg.MODEL.RPN.IOU_THRESHOLDS, cfg.MODEL.RPN.IOU_LABELS,
                                      allow_low_quality_matches=True),
            "head": head,
        }

    def forward(self, images, features: Dict[str, torch.Tensor], gt_instances=None, head_outputs=None, padded=False):
        """head_outputs: (logits, deltas) when the head already ran inside the captured dense graph.  padded=True (eval,
        device path): every image gets all post-NMS slots, empty ones with objectness -inf, and no host sync happens
        here; the caller's box predictor drops them (FastRCNNOutputs.inference)."""
        feats = [features[f] for f in self.in_features]
        grid_sizes = [(f.shape[1], f.shape[2]) for f in feats]                     # NHWC
        anchors = self.anchor_generator(grid_sizes, feats[0].device)
        if head_outputs is None:
            pred_objectness_logits, pred_anchor_deltas = self.rpn_head(feats)
        else:
            pred_objectness_logits, pred_anchor_deltas = head_outputs
        if self.training:
            assert gt_instances is not None, "RPN requires gt_instances in training!"
            # training runs on the static-shape path (modeling/dense_train.py: rpn_label_and_sample / rpn_losses on fused
            # kernels).  The per-image list formulation (label_and_sample_anchors / losses, rpn.py:41-354 of the reference)
            # is test infrastructure: oracle/list_path.py attaches it (oracle.list_path.install).
            if not hasattr(self, "losses"):
                raise RuntimeError("RPN.forward(training) on instance lists: use the static-shape path (model.dense_train "
                                   "= True); the list formulation is oracle/list_path.py")
            gt_labels, gt_boxes = self.label_and_sample_anchors(anchors, gt_instances)
            losses = self.losses(anchors, pred_objectness_logits, gt_labels, pred_anchor_deltas, gt_boxes)
        else:
            losses = {}
        proposals = self.predict_proposals(anchors, pred_objectness_logits, pred_anchor_deltas, images.image_sizes,
                                           padded=padded and not self.training)
        return proposals, losses

    @torch.no_grad()
    def predict_proposals(self, anchors, pred_objectness_logits, pred_anchor_deltas, image_sizes, padded=False):
        if pred_anchor_deltas[0].is_cuda and hasattr(ops, "rpn_decode_select"):
            # device path: ONE batched top-k over the levels, only the candidates are decoded (fused kernel), grouped
            # NMS, post-NMS top-k; a single host sync for the per-image proposal counts
            from ..dense_train import rpn_proposals_padded
            boxes, scores = rpn_proposals_padded(self, torch.cat([a.tensor for a in anchors]), pred_objectness_logits,
                                                 torch.cat([d.detach() for d in pred_anchor_deltas], 1), image_sizes,
                                                 training=self.training)
            if padded:
                return [Instances(image_sizes[i], proposal_boxes=Boxes(boxes[i]), objectness_logits=scores[i])
                        for i in range(len(image_sizes))]
            counts = torch.isfinite(scores).sum(1).tolist()
            results = []
            for i, n in enumerate(counts):
                res = Instances(image_sizes[i])
                res.proposal_boxes = Boxes(boxes[i, :n])
                res.objectness_logits = scores[i, :n]
                results.append(res)
            return results
        pred_proposals = self._decode_proposals(anchors, pred_anchor_deltas)
        return find_top_rpn_proposals(pred_proposals, [t.detach() for t in pred_objectness_logits], image_sizes,
                                      self.nms_thresh, self.pre_nms_topk[self.training],
                                      self.post_nms_topk[self.training], self.min_box_size, self.training)

    def _decode_proposals(self, anchors, pred_anchor_deltas):
        N = pred_anchor_deltas[0].shape[0]
        proposals = []
        for anchors_i, pred_anchor_deltas_i in zip(anchors, pred_anchor_deltas):
            B = anchors_i.tensor.size(1)
            pred_anchor_deltas_i = pred_anchor_deltas_i.detach().reshape(-1, B)
            anchors_e = anchors_i.tensor.unsqueeze(0).expand(N, -1, -1).reshape(-1, B)
            proposals_i = self.box2box_transform.apply_deltas(pred_anchor_deltas_i, anchors_e)
            proposals.append(proposals_i.view(N, -1, B))
        return proposals


@PROPOSAL_GENERATOR_REGISTRY.register()
class RPNWithIgnore(RPN):
    def __init__(self, *, ignore_thresh: float = 0.5, objectness_uncertainty: str = "none", **kwargs):
        super().__init__(**kwargs)
        self.ignore_thresh = ignore_thresh
        self.objectness_uncertainty = objectness_uncertainty

    @classmethod
    def from_config(cls, cfg, input_shape: Dict[str, ShapeSpec]):
        ret = super().from_config(cfg, input_shape)
        ret["ignore_thresh"] = cfg.MODEL.RPN.IGNORE_THRESHOLD
        ret["objectness_uncertainty"] = cfg.MODEL.RPN.OBJECTNESS_UNCERTAINTY
        return ret

    @torch.no_grad()
    def label_and_sample_anchors(self, anchors: List[Boxes], gt_instances: List[Instances]):
        """rpn.py:41-110 under its reference name and signature, computed by the fused kernels of the static-shape path
        (dense_train.rpn_label_and_sample: IoU matching with the forced best anchor per object, IoU-weighted sampling without
        replacement, ignore regions by IoA): -> (list of (A,) int8 labels in {-1, 0, 1}, list of (A,4) matched gt boxes)."""
        from ..dense_train import GTBatch, rpn_label_and_sample, matched_boxes
        A = Boxes.cat(anchors).tensor
        gt = GTBatch(gt_instances, A.device)
        labels, midx, _ = rpn_label_and_sample(self, A, gt)
        boxes = matched_boxes(gt, midx)
        return [l.to(torch.int8) for l in labels], [b for b in boxes]

def _construct(cls, cfg, *args, **kwargs):
    return cls(**cls.from_config(cfg, *args, **kwargs))


def build_proposal_generator(cfg, input_shape):
    name = cfg.MODEL.PROPOSAL_GENERATOR.NAME
    if name == "PrecomputedProposals":
        return None
    return _construct(PROPOSAL_GENERATOR_REGISTRY.get(name), cfg, input_shape)
