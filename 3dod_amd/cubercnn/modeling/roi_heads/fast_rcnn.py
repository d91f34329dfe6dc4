"""2D box predictor -- cubercnn/modeling/roi_heads/fast_rcnn.py:57-260 (FastRCNNOutputs) on top of
detectron2's FastRCNNOutputLayers [third-party, restated]."""
from typing import List, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ....d2lite import Boxes, Instances, Box2BoxTransform, cat, get_event_storage
from .... import hipops as ops

bf16 = torch.bfloat16


def batched_nms(boxes, scores, idxs, iou_threshold):
    """torchvision.ops.batched_nms semantics [third-party] on cr_nms_grouped: returns kept indices sorted by
    descending score."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    order = torch.argsort(scores, descending=True, stable=True)
    b, cls = boxes[order], idxs[order]
    # one group per class, padded to the largest class
    uniq, inv, counts = torch.unique(cls, return_inverse=True, return_counts=True)
    G, maxn = uniq.numel(), int(counts.max())
    srt = torch.argsort(inv, stable=True)                      # group-major, score-descending inside a group
    starts = torch.cumsum(counts, 0) - counts
    pos = torch.arange(srt.numel(), device=boxes.device) - starts[inv[srt]]
    pad = b.new_zeros((G, maxn, 4))
    pad[inv[srt], pos] = b[srt]
    keep = ops.nms_grouped(pad, counts.to(torch.int32), iou_threshold)
    kept_sorted_pos = srt[keep[inv[srt], pos]]
    kept = order[torch.sort(kept_sorted_pos)[0]]
    return kept


def fast_rcnn_inference_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk_per_image):
    """fast_rcnn.py:57-116."""
    valid_mask = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    if not valid_mask.all():
        boxes = boxes[valid_mask]
        scores = scores[valid_mask]
    scores = scores[:, :-1]
    num_bbox_reg_classes = boxes.shape[1] // 4
    boxes = Boxes(boxes.reshape(-1, 4))
    boxes.clip(image_shape)
    boxes = boxes.tensor.view(-1, num_bbox_reg_classes, 4)
    filter_mask = scores > score_thresh
    filter_inds = filter_mask.nonzero()
    if num_bbox_reg_classes == 1:
        boxes = boxes[filter_inds[:, 0], 0]
    else:
        boxes = boxes[filter_mask]
    scores_full = scores[filter_inds[:, 0]]
    scores = scores[filter_mask]
    keep = batched_nms(boxes, scores, filter_inds[:, 1], nms_thresh)
    if topk_per_image >= 0:
        keep = keep[:topk_per_image]
    boxes, scores, filter_inds, scores_full = boxes[keep], scores[keep], filter_inds[keep], scores_full[keep]
    result = Instances(image_shape)
    result.pred_boxes = Boxes(boxes)
    result.scores = scores
    result.scores_full = scores_full
    result.pred_classes = filter_inds[:, 1]
    return result, filter_inds[:, 0]


def fast_rcnn_inference(boxes, scores, image_shapes, score_thresh, nms_thresh, topk_per_image):
    """fast_rcnn.py:57-116 for the whole batch at once: the reference filters / NMSes / truncates image by image; here
    the candidates of all images go through ONE grouped NMS (group = image x class) and one per-image top-k, with two
    host syncs per batch instead of several per image.  Same result as fast_rcnn_inference_single_image per image."""
    if len(boxes) <= 1:
        res = [fast_rcnn_inference_single_image(b, s, shp, score_thresh, nms_thresh, topk_per_image)
               for s, b, shp in zip(scores, boxes, image_shapes)]
        return [x[0] for x in res], [x[1] for x in res]
    dev = boxes[0].device
    sizes = [int(b.shape[0]) for b in boxes]
    B, S = torch.cat(boxes), torch.cat(scores)
    n_img, K = len(boxes), S.shape[1] - 1
    img = torch.repeat_interleave(torch.arange(n_img, device=dev), torch.tensor(sizes, device=dev))
    start = torch.tensor([sum(sizes[:i]) for i in range(n_img)], device=dev)
    valid = torch.isfinite(B).all(dim=1) & torch.isfinite(S).all(dim=1)
    S = S[:, :-1]
    nreg = B.shape[1] // 4
    hw = torch.tensor([[s[1], s[0], s[1], s[0]] for s in image_shapes], dtype=B.dtype, device=dev)[img]      # (R,4)
    Bc = torch.minimum(B.view(-1, nreg, 4).clamp(min=0), hw[:, None, :])                                         # Boxes.clip
    mask = (S > score_thresh) & valid[:, None]
    inds = mask.nonzero()                                                   # (n, 2) = (proposal row, class); host sync 1
    bsel = Bc[inds[:, 0], 0] if nreg == 1 else Bc[mask]
    ssel, sfull = S[mask], S[inds[:, 0]]
    grp = img[inds[:, 0]] * K + inds[:, 1]
    keep = batched_nms(bsel, ssel, grp, nms_thresh)                        # descending score over the whole batch
    kimg = img[inds[keep, 0]]
    order = torch.argsort(kimg, stable=True)                               # image-major, score-descending inside
    keep, kimg = keep[order], kimg[order]
    counts = torch.bincount(kimg, minlength=n_img)
    if topk_per_image >= 0:
        first = torch.cumsum(counts, 0) - counts
        rank = torch.arange(keep.numel(), device=dev) - first[kimg]
        sel = rank < topk_per_image
        keep, kimg = keep[sel], kimg[sel]
        counts = counts.clamp(max=topk_per_image)
    splits = counts.tolist()                                                # host sync 2
    results, kept_rows = [], []
    rows = inds[keep, 0] - start[kimg]
    for i, (kb, ks, kf, kc, kr) in enumerate(zip(bsel[keep].split(splits), ssel[keep].split(splits),
                                                 sfull[keep].split(splits), inds[keep, 1].split(splits),
                                                 rows.split(splits))):
        r = Instances(image_shapes[i])
        r.pred_boxes = Boxes(kb)
        r.scores = ks
        r.scores_full = kf
        r.pred_classes = kc
        results.append(r)
        kept_rows.append(kr)
    return results, kept_rows


class FastRCNNOutputLayers(nn.Module):
    """detectron2 FastRCNNOutputLayers [third-party, restated]: cls_score (K+1) and bbox_pred (K*4) linears."""

    def __init__(self, input_size, *, box2box_transform, num_classes, test_score_thresh=0.0, test_nms_thresh=0.5,
                 test_topk_per_image=100, cls_agnostic_bbox_reg=False, smooth_l1_beta=0.0,
                 box_reg_loss_type="smooth_l1", loss_weight=1.0):
        super().__init__()
        self.num_classes = num_classes
        self.cls_score = nn.Linear(input_size, num_classes + 1)
        num_bbox_reg_classes = 1 if cls_agnostic_bbox_reg else num_classes
        self.bbox_pred = nn.Linear(input_size, num_bbox_reg_classes * 4)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        for l in [self.cls_score, self.bbox_pred]:
            nn.init.constant_(l.bias, 0)
        self.box2box_transform = box2box_transform
        self.smooth_l1_beta = smooth_l1_beta
        self.test_score_thresh = test_score_thresh
        self.test_nms_thresh = test_nms_thresh
        self.test_topk_per_image = test_topk_per_image
        self.box_reg_loss_type = box_reg_loss_type
        if isinstance(loss_weight, float):
            loss_weight = {"loss_cls": loss_weight, "loss_box_reg": loss_weight}
        self.loss_weight = loss_weight

    def forward(self, x):
        if x.dim() > 2:
            x = torch.flatten(x, start_dim=1)
        # both predictors as one GEMM (rows padded to the MFMA tile inside ops.linear_cat)
        y, offs = ops.linear_cat(x, [self.cls_score.weight, self.bbox_pred.weight], [self.cls_score.bias, self.bbox_pred.bias])
        return y[:, offs[0]:offs[1]].float(), y[:, offs[1]:offs[2]].float()

    def predict_boxes_for_gt_classes(self, predictions, proposals):
        if not len(proposals):
            return []
        scores, proposal_deltas = predictions
        proposal_boxes = cat([p.proposal_boxes.tensor for p in proposals], dim=0)
        N, B = proposal_boxes.shape
        predict_boxes = self.box2box_transform.apply_deltas(proposal_deltas, proposal_boxes)
        K = predict_boxes.shape[1] // B
        if K > 1:
            gt_classes = torch.cat([p.gt_classes for p in proposals], dim=0)
            gt_classes = gt_classes.clamp_(0, K - 1)
            predict_boxes = predict_boxes.view(N, K, B)[torch.arange(N, dtype=torch.long, device=predict_boxes.device),
                                                          gt_classes]
        num_prop_per_image = [len(p) for p in proposals]
        return predict_boxes.split(num_prop_per_image)

    def predict_boxes(self, predictions, proposals):
        if not len(proposals):
            return []
        _, proposal_deltas = predictions
        num_prop_per_image = [len(p) for p in proposals]
        proposal_boxes = cat([p.proposal_boxes.tensor for p in proposals], dim=0)
        predict_boxes = self.box2box_transform.apply_deltas(proposal_deltas, proposal_boxes)
        return predict_boxes.split(num_prop_per_image)

    def predict_probs(self, predictions, proposals):
        scores, _ = predictions
        num_inst_per_image = [len(p) for p in proposals]
        probs = F.softmax(scores, dim=-1)
        return probs.split(num_inst_per_image, dim=0)


class FastRCNNOutputs(FastRCNNOutputLayers):
    def __init__(self, cfg, input_shape):
        input_size = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        super().__init__(
            input_size,
            box2box_transform=Box2BoxTransform(weights=cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS),
            num_classes=cfg.MODEL.ROI_HEADS.NUM_CLASSES,
            cls_agnostic_bbox_reg=cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG,
            smooth_l1_beta=cfg.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA,
            test_score_thresh=cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
            test_nms_thresh=cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST,
            test_topk_per_image=cfg.TEST.DETECTIONS_PER_IMAGE,
            box_reg_loss_type=cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE,
            loss_weight={"loss_box_reg": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT},
        )

    def inference(self, predictions, proposals):
        boxes = self.predict_boxes(predictions, proposals)
        if len(proposals) and proposals[0].has("objectness_logits"):
            # padded proposal slots of the RPN's sync-free eval path (objectness -inf) must not become detections: their
            # scores turn NaN, which fast_rcnn_inference filters like any non-finite prediction
            real = torch.isfinite(cat([p.objectness_logits for p in proposals], dim=0))
            probs = F.softmax(predictions[0], dim=-1)
            probs = torch.where(real[:, None], probs, torch.full((), float("nan"), dtype=probs.dtype, device=probs.device))
            scores = probs.split([len(p) for p in proposals], dim=0)
        else:
            scores = self.predict_probs(predictions, proposals)
        image_shapes = [x.image_size for x in proposals]
        return fast_rcnn_inference(boxes, scores, image_shapes, self.test_score_thresh, self.test_nms_thresh,
                                   self.test_topk_per_image)

    def losses(self, predictions, proposals):
        """fast_rcnn.py:145-194."""
        scores, proposal_deltas = predictions
        gt_classes = cat([p.gt_classes for p in proposals], dim=0) if len(proposals) else torch.empty(0)
        if len(proposals):
            proposal_boxes = cat([p.proposal_boxes.tensor for p in proposals], dim=0)
            gt_boxes = cat([(p.gt_boxes if p.has("gt_boxes") else p.proposal_boxes).tensor for p in proposals], dim=0)
        else:
            proposal_boxes = gt_boxes = torch.empty((0, 4), device=proposal_deltas.device)
        normalize_factor = max(gt_classes.numel(), 1.0)
        # detectron2 cross_entropy wrapper: 0 for empty input
        loss_cls = F.cross_entropy(scores, gt_classes, reduction="mean") if gt_classes.numel() else scores.sum() * 0.0
        loss_box_reg = self.box_reg_loss(proposal_boxes, gt_boxes, proposal_deltas, gt_classes, reduction="none")
        loss_box_reg = loss_box_reg.sum() / normalize_factor
        losses = {"BoxHead/loss_cls": loss_cls, "BoxHead/loss_box_reg": loss_box_reg}
        return {k: v * self.loss_weight.get(k, 1.0) for k, v in losses.items()}

    def box_reg_loss(self, proposal_boxes, gt_boxes, pred_deltas, gt_classes, reduction='mean'):
        """fast_rcnn.py:196-260 (smooth_l1 with beta, reduction 'none' or 'mean')."""
        box_dim = proposal_boxes.shape[1]
        fg_inds = ((gt_classes >= 0) & (gt_classes < self.num_classes)).nonzero(as_tuple=True)[0]
        if pred_deltas.shape[1] == box_dim:
            fg_pred_deltas = pred_deltas[fg_inds]
        else:
            fg_pred_deltas = pred_deltas.view(-1, self.num_classes, box_dim)[fg_inds, gt_classes[fg_inds]]
        if self.box_reg_loss_type != "smooth_l1":
            raise ValueError(f"Invalid bbox reg loss type '{self.box_reg_loss_type}'")
        gt_pred_deltas = self.box2box_transform.get_deltas(proposal_boxes[fg_inds], gt_boxes[fg_inds])
        nd = torch.abs(fg_pred_deltas - gt_pred_deltas)
        if self.smooth_l1_beta < 1e-5:
            loss = nd
        else:
            loss = torch.where(nd < self.smooth_l1_beta, 0.5 * nd ** 2 / self.smooth_l1_beta,
                               nd - 0.5 * self.smooth_l1_beta)
        if reduction == 'mean':
            return loss.sum() / max(gt_classes.numel(), 1.0)
        elif reduction == 'none':
            return loss
        raise ValueError(f"Invalid bbox reg reduction type '{reduction}'")
