"""Static-shape ("dense") training path of RCNN3D: the same labelling / sampling / loss rules as the reference
(cubercnn/modeling/proposal_generator/rpn.py:41-328, roi_heads/roi_heads.py:2737-2840, fast_rcnn.py:145-260 and
the detectron2 Matcher / subsample rules they build on) written over PADDED, fixed-shape tensors with validity
masks instead of data-dependent shapes.

Why: every boolean index / nonzero() / .tolist() of the reference formulation is a host<->device sync; a Cube R-CNN
step has ~60 of them, which leaves the GPU idle while Python catches up (profiles/: 23 ms of kernels in a 41 ms
step).  With fixed shapes nothing syncs, the host runs ahead of the device, and the step becomes capturable.

Sampling rule equivalences (no semantic change):
  * torch.multinomial(w, k) without replacement == the k largest keys w / E, E ~ Exp(1) (that is literally ATen's
    implementation), so "min(#candidates, k) IoU-weighted samples" == top-k keys with non-candidates keyed 0 and
    flagged invalid;
  * "fill the rest with negatives" == the first (num_samples - #positives taken) of the top negatives.
Losses are masked sums divided by the same (device-side) counts the reference divides by.
"""
import numpy as np
import torch
import torch.nn.functional as F

from ...d2lite import Boxes, get_event_storage
from ... import hipops as ops
from ..util import math_util as util

NEG = -1.0


class GTBatch:
    """ground truth of a batch padded to G objects per image."""

    def __init__(self, gt_instances, device, G=None):
        B = len(gt_instances)
        need = max(1, max(len(g) for g in gt_instances))
        G = need if G is None else G
        assert need <= G, f"{need} objects in an image but the static GT buffer holds {G}"
        device = torch.device(device)
        if device.type == "cuda" and hasattr(ops, "gt_pack") and B <= 32:
            self.boxes = torch.empty((B, G, 4), device=device)
            self.classes = torch.empty((B, G), dtype=torch.int64, device=device)    # -2 = padding, -1 = ignore
            self.boxes3D = torch.empty((B, G, 9), device=device)
            self.poses = torch.empty((B, G, 3, 3), device=device)
            ops.gt_pack(gt_instances, G, self.boxes, self.classes, self.boxes3D, self.poses)        # one launch
            return
        self.boxes = torch.zeros((B, G, 4), device=device)
        self.classes = torch.full((B, G), -2, dtype=torch.int64, device=device)     # -2 = padding, -1 = ignore
        self.boxes3D = torch.zeros((B, G, 9), device=device)
        self.poses = torch.eye(3, device=device).expand(B, G, 3, 3).clone()
        for i, g in enumerate(gt_instances):
            n = len(g)
            if n:
                self.boxes[i, :n] = g.gt_boxes.tensor
                self.classes[i, :n] = g.gt_classes
                if g.has("gt_boxes3D"):
                    self.boxes3D[i, :n] = g.gt_boxes3D
                    self.poses[i, :n] = g.gt_poses

    @property
    def valid(self):
        return self.classes >= 0

    @property
    def ignore(self):
        return self.classes == -1

    def refill(self, gt_instances):
        """in-place refresh of a static (graph-captured) buffer set from a new batch: one launch on the GPU"""
        B, G = self.classes.shape
        if self.boxes.is_cuda and hasattr(ops, "gt_pack") and B <= 32:
            assert len(gt_instances) == B and max(len(g) for g in gt_instances) <= G
            ops.gt_pack(gt_instances, G, self.boxes, self.classes, self.boxes3D, self.poses)
        else:
            self.copy_from(GTBatch(gt_instances, self.boxes.device, G=G))

    def copy_from(self, other):
        """in-place refresh of a static (graph-captured) buffer set."""
        self.boxes.copy_(other.boxes)
        self.classes.copy_(other.classes)
        self.boxes3D.copy_(other.boxes3D)
        self.poses.copy_(other.poses)


def camera_meta(rh, Ks, im_scales_ratio, im_dims, device):
    """per-image camera constants (B,5) = [fx,fy,cx,cy of K/ratio, virtual_to_real] (roi_heads.py:2285-2315) made on
    the host and shipped with ONE pinned, non-blocking copy (no sync)."""
    rows = []
    for k, r, d in zip(Ks, im_scales_ratio, im_dims):
        k = torch.as_tensor(k, dtype=torch.float32)
        v2r_i = util.compute_virtual_scale_from_focal_spaces(float(k[1, 1]), float(d[0]) * float(r), rh.virtual_focal,
                                                             float(d[0])) if rh.virtual_depth else 1.0
        rows.append([float(k[0, 0]) / r, float(k[1, 1]) / r, float(k[0, 2]) / r, float(k[1, 2]) / r, float(v2r_i)])
    meta = torch.tensor(rows, dtype=torch.float32)
    return (meta.pin_memory() if device.type == "cuda" else meta).to(device, non_blocking=True)


# ------------------------------------------------------------------------------------------- RPN
def rpn_label_and_sample(rpn, anchors, gt: GTBatch):
    """RPNWithIgnore.label_and_sample_anchors (rpn.py:41-110) for the whole batch.  anchors (A,4).
    Returns labels (B,A) int32 in {-1,0,1}, matched gt index (B,A) int32, matched_ious (B,A).
    Five launches + two top-k: match, label + keys, scatter (ops.box_match / rpn_label / rpn_scatter)."""
    B, A = gt.boxes.shape[0], anchors.shape[0]
    max_iou, midx, ioa, best = ops.box_match(anchors, gt.boxes, gt.classes, want_best=True)
    lo, hi = rpn.anchor_matcher.thresholds[1], rpn.anchor_matcher.thresholds[2]
    expo = torch.empty((2, B, A), device=anchors.device).exponential_(1.0)
    # keys = (IoU + eps) / Exp(1) on the positive / negative candidates: their top-k is the IoU-weighted multinomial
    # sampling without replacement of rpn.py:275-328
    _, out, matched_ious, keys = ops.rpn_label(anchors, gt.boxes, gt.classes, max_iou, best, expo, lo, hi,
                                               rpn.anchor_matcher.labels, 1e-4)
    n_s = rpn.batch_size_per_image
    k_pos = int(n_s * rpn.positive_fraction)
    # one top-k launch for both key sets (positives need the first k_pos of theirs; sorted output)
    kk = min(max(k_pos, n_s), A)
    tkey, tidx = ops.topk(keys.view(2 * B, A), kk)
    pkey, pidx = tkey[:B, :min(k_pos, A)], tidx[:B, :min(k_pos, A)]
    nkey, nidx = tkey[B:, :min(n_s, A)], tidx[B:, :min(n_s, A)]
    # rpn.py:75 (forced arg-max anchors) is already in `out`; rpn.py:93-104 (ignore regions) inside the scatter
    ops.rpn_scatter(out, pidx, pkey, nidx, nkey, n_s, ioa, rpn.ignore_thresh)
    return out, midx, matched_ious


def matched_boxes(gt: GTBatch, midx):
    """(B,A,4) matched gt box per anchor (zeros for images without objects), as the reference's per-image lists."""
    has_gt = gt.valid.any(1)
    return torch.gather(gt.boxes, 1, midx.long()[:, :, None].expand(-1, -1, 4)) * has_gt[:, None, None]


def rpn_losses(rpn, anchors, logits, deltas, labels, midx, gt: GTBatch):
    """RPNWithIgnore.losses + _dense_box_regression_loss_with_uncertainty (rpn.py:129-273), "IoUness" objectness.
    logits (B,A), deltas (B,A,4); one fused kernel computes both sums, the logging counters and the gradients."""
    assert rpn.objectness_uncertainty.lower() != "none" and rpn.box_reg_loss_type == "smooth_l1" and rpn.smooth_l1_beta < 1e-5
    B, A = labels.shape
    loss_conf, loss_loc, sums = ops.rpn_loss(logits, deltas, anchors, labels, midx, gt.boxes, rpn.box2box_transform.weights)
    storage = get_event_storage()
    with torch.no_grad():
        # the four logging values as ONE vector expression: [npos, nneg, conf_pos, conf_neg] / [B, B, max(npos,1), max(rest,1)]
        cnt = sums[2:4] / B                                              # anchors per image: positive, negative
        conf = sums[4:6] / torch.stack([sums[2], B * A - sums[2]]).clamp(min=1)
        storage.put_scalar("rpn/num_pos_anchors", cnt[0])
        storage.put_scalar("rpn/num_neg_anchors", cnt[1])
        storage.put_scalar("rpn/conf_pos_anchors", conf[0])
        storage.put_scalar("rpn/conf_neg_anchors", conf[1])
    normalizer = rpn.batch_size_per_image * B
    return {"rpn/cls": loss_conf * (rpn.loss_weight.get("rpn/cls", 1.0) / normalizer),
            "rpn/loc": loss_loc * (rpn.loss_weight.get("rpn/loc", 1.0) / normalizer)}


_PCONST = {}
_WVEC = {}


class RawRPNOutputs:
    """the RPN head's per-level outputs y_l (B,H,W,16) as they leave the one 16-channel convolution (dense-region graph)"""
    __slots__ = ("ys",)

    def __init__(self, ys):
        self.ys = list(ys)


def rpn_tensors(rpn, feats, head_outputs):
    """-> (logits (B,Atot), deltas (B,Atot,4), per-level logits padded with -inf (B,L,amax) or None, anchors per level).
    head_outputs: None (run the head), RawRPNOutputs, or the (logits per level, deltas per level) lists of RPN.forward."""
    A = rpn.rpn_head.num_anchors
    if head_outputs is None or isinstance(head_outputs, RawRPNOutputs):
        ys = head_outputs.ys if head_outputs is not None else rpn.rpn_head.forward_raw(feats)
        if len(ys) == 1 and len(feats) > 1:                           # the levels stacked in one map (StandardRPNHead.forward_raw)
            cells = [int(f.shape[1] * f.shape[2]) for f in feats]
            logits, deltas, padded = ops.rpn_unpack_stacked(ys[0], cells, feats[0].shape[0], A)
            return logits, deltas, padded, [c * A for c in cells]
        logits, deltas, padded = ops.rpn_unpack(ys, A)               # one launch each way
        return logits, deltas, padded, [int(y.shape[1] * y.shape[2] * A) for y in ys]
    logits_lv, deltas_lv = head_outputs
    return torch.cat(logits_lv, 1), torch.cat(deltas_lv, 1), None, [int(t.shape[1]) for t in logits_lv]


def rpn_proposals_padded(rpn, anchors, logits_per_level, deltas, image_sizes, training=True, padded=None, sizes=None):
    """RPN.predict_proposals / find_top_rpn_proposals (detectron2 [third-party], restated in proposal_generator/rpn.py)
    with a padded result: boxes (B,K,4), scores (B,K) (-inf = empty slot).  anchors (A,4) and deltas (B,A,4) are the
    level-concatenated tensors.  Only the per-level top-k candidates are decoded (one fused launch: decode, clip,
    validity), instead of decoding every anchor of every level first."""
    with torch.no_grad():
        B = deltas.shape[0]
        dev = deltas.device
        if sizes is None:
            sizes = [int(t.shape[1]) for t in logits_per_level]
        ks = [min(n, rpn.pre_nms_topk[training]) for n in sizes]
        maxn, L = max(ks), len(ks)
        ckey = (tuple(tuple(s) for s in image_sizes), tuple(sizes), tuple(ks), str(dev))
        cached = _PCONST.get(ckey)
        if cached is None:       # constants of the configuration: made once, never inside a captured region
            cached = [torch.tensor([[float(s[0]), float(s[1])] for s in image_sizes], dtype=torch.float32, device=dev),
                      torch.tensor(ks, dtype=torch.int32, device=dev).repeat(B),
                      torch.tensor([sum(sizes[:l]) for l in range(len(sizes))], dtype=torch.int64, device=dev).view(1, -1, 1),
                      torch.full((), -1, dtype=torch.int64, device=dev)]
            _PCONST[ckey] = cached
        img_hw, counts = cached[0], cached[1]
        # ONE top-k for all levels: logits padded to the largest level with -inf (a per-level call costs as much as
        # this single batched one: the select is one workgroup per row either way)
        amax = max(sizes)
        if padded is None:
            padded = torch.full((B, L, amax), float("-inf"), dtype=torch.float32, device=dev)
            for l, lg in enumerate(logits_per_level):
                padded[:, l, :sizes[l]] = lg.detach()
        kq = min(maxn, amax)
        scores, idx = ops.topk(padded.view(B * L, amax), kq)                  # sorted descending
        scores, idx = scores.view(B, L, kq), idx.view(B, L, kq)
        # (top-k values are finite logits or the -inf padding: one compare instead of isfinite's four kernels)
        idx = torch.where(scores > -3.0e38, idx + cached[2], cached[3])
        t = rpn.box2box_transform
        boxes, nms_boxes, valid = ops.rpn_decode_select(anchors, deltas.detach(), idx.view(B, -1), scores.view(B, -1),
                                                        t.weights, t.scale_clamp, img_hw, rpn.min_box_size)
        keep = ops.nms_grouped(nms_boxes.view(B * L, maxn, 4), counts, rpn.nms_thresh).view(B, -1) & valid
        flat_scores = torch.where(keep, scores.view(B, -1), torch.full((), float("-inf"), device=dev))
        k_post = min(rpn.post_nms_topk[training], flat_scores.shape[1])
        top_scores, top_idx = ops.topk(flat_scores, k_post)
        return torch.gather(boxes, 1, top_idx[:, :, None].expand(-1, -1, 4)), top_scores


# ------------------------------------------------------------------------------------------- ROI heads
@torch.no_grad()
def roi_label_and_sample(rh, prop_boxes, prop_scores, gt: GTBatch):
    """ROIHeads3D.label_and_sample_proposals (roi_heads.py:2773-2840): append GT, match, ignore, IoU-weighted
    sampling of <= 25 % foreground + background up to 512.  Output slots: [0,k_fg) foreground picks, then background.
    Returns dict(boxes (B,S,4), valid (B,S), classes (B,S) with num_classes = background, gt_idx (B,S), k_fg)."""
    dev = prop_boxes.device
    B = prop_boxes.shape[0]
    K = rh.num_classes
    prop_valid = prop_scores > -3.0e38                  # proposal slots are finite scores or -inf (empty)
    if rh.proposal_append_gt:
        boxes = torch.cat([prop_boxes, gt.boxes], 1)
        valid = torch.cat([prop_valid, gt.valid], 1)
    else:
        boxes, valid = prop_boxes, prop_valid
    R = boxes.shape[1]
    max_iou, midx, ioa, _ = ops.box_match(boxes, gt.boxes, gt.classes)
    expo = torch.empty((2, B, R), device=dev).exponential_(1.0)
    cls, _, keys = ops.roi_label(max_iou, midx, ioa, valid, gt.classes, expo, K, rh.proposal_matcher.thresholds[1],
                                 rh.ignore_thresh, 1e-4)
    n_s = rh.batch_size_per_image
    k_fg = min(int(n_s * rh.positive_fraction), R)
    kk = min(max(k_fg, n_s), R)                       # one top-k launch for the foreground and background keys
    tkey, tidx = ops.topk(keys.view(2 * B, R), kk)
    fkey, fidx = tkey[:B, :k_fg], tidx[:B, :k_fg]
    bkey, bidx = tkey[B:, :min(n_s, R)], tidx[B:, :min(n_s, R)]
    # at most n_s picks are valid; the stable compaction keeps "foreground first" (the k_fg leading slots hold every
    # valid foreground pick)
    s_boxes, s_valid, s_cls, s_gt, counts = ops.roi_compact(fidx, fkey, bidx, bkey, n_s, boxes, cls, midx)
    storage = get_event_storage()
    cm = counts.float().mean(0)
    storage.put_scalar("roi_head/num_fg_samples", cm[0])
    storage.put_scalar("roi_head/num_bg_samples", cm[1])
    return {"boxes": s_boxes, "valid": s_valid, "classes": s_cls, "gt_idx": s_gt, "k_fg": k_fg}


_BIDX = {}


def _rois(boxes):
    """(B,n,4) -> (B*n,5) [image index, x1,y1,x2,y2] (the index column is a cached constant)."""
    B, n = boxes.shape[:2]
    key = (B, n, str(boxes.device))
    col = _BIDX.get(key)
    if col is None:
        col = _BIDX[key] = torch.arange(B, device=boxes.device, dtype=boxes.dtype).repeat_interleave(n)[:, None]
    return torch.cat([col, boxes.reshape(B * n, 4)], 1)


def pool_roi_features(rh, features, samp):
    """ROIAlign for the box head (all S slots) and the cube head (the k_fg foreground slots, optionally rescaled boxes,
    roi_heads.py:2217-2235) in ONE launch when both poolers share their configuration: one forward kernel, and one
    backward accumulation pyramid instead of two (zero-fill, atomics, cast and the autograd add of the two results)."""
    B, S = samp["valid"].shape
    kf = samp["k_fg"]
    cb = samp["boxes"][:, :kf]
    if rh.scale_roi_boxes > 0:           # (the reference uses the width for the height as well)
        ctr = (cb[..., :2] + cb[..., 2:]) * 0.5
        half = (cb[..., 2:3] - cb[..., 0:1]) * (0.5 * rh.scale_roi_boxes)
        cb = torch.cat([ctr - half, ctr + half], -1)
    bp, cp = rh.box_pooler, rh.cube_pooler
    same = rh.loss_w_3d > 0 and tuple(rh.box_in_features) == tuple(rh.in_features) and bp.scales == cp.scales and \
        bp.output_size == cp.output_size
    if same and not rh.scale_roi_boxes > 0 and hasattr(ops, "shared_prefix"):
        # the 3D head's RoIs ARE the first kf RoIs of the box head: pool once, share (forward and backward)
        out = ops.roi_align_pyramid([features[f] for f in rh.in_features], _rois(samp["boxes"]), bp.scales, bp.output_size)
        return ops.shared_prefix(out, B, S, kf)
    if same:
        feats = [features[f] for f in rh.in_features]
        out = ops.roi_align_pyramid(feats, torch.cat([_rois(samp["boxes"]), _rois(cb)], 0), bp.scales, bp.output_size)
        return out[:B * S], out[B * S:]
    box = ops.roi_align_pyramid([features[f] for f in rh.box_in_features], _rois(samp["boxes"]), bp.scales, bp.output_size)
    cube = ops.roi_align_pyramid([features[f] for f in rh.in_features], _rois(cb), cp.scales, cp.output_size) \
        if rh.loss_w_3d > 0 else None
    return box, cube


def box_head_losses(rh, features, samp, gt: GTBatch, pooled=None):
    """_forward_box in training (roi_heads.py:2160-2204) + FastRCNNOutputs.losses (fast_rcnn.py:145-194) on the padded
    sample.  Returns (losses, pred_boxes (B,S,4) for the sampled classes)."""
    B, S = samp["valid"].shape
    if pooled is None:
        pooled = pool_roi_features(rh, features, samp)[0]
    box_features = rh.box_head(pooled)
    scores, deltas = rh.box_predictor(box_features)                                    # (B*S,K+1), (B*S,K*4)
    assert rh.box_predictor.smooth_l1_beta < 1e-5
    t = rh.box_predictor.box2box_transform
    sum_ce, sum_l1, sums, pred = ops.box_loss(scores, deltas, samp["valid"], samp["classes"], samp["boxes"], samp["gt_idx"],
                                              gt.boxes, t.weights, t.scale_clamp)
    inv = sums[2].clamp(min=1).reciprocal()
    lw = rh.box_predictor.loss_weight
    w_cls, w_reg = lw.get("BoxHead/loss_cls", 1.0), lw.get("BoxHead/loss_box_reg", 1.0)
    losses = {"BoxHead/loss_cls": sum_ce * (inv if w_cls == 1.0 else inv * w_cls),
              "BoxHead/loss_box_reg": sum_l1 * (inv if w_reg == 1.0 else inv * w_reg)}
    return losses, pred.view(B, S, 4)


def cube_head_losses(rh, features, samp, pred_boxes, gt: GTBatch, meta, pooled=None):
    """_forward_cube in training (roi_heads.py:2237-2679) on the k_fg foreground slots of every image.  Empty slots
    are excluded from the reductions (safely_reduce_losses, roi_heads.py:2843-2851) by their validity flag."""
    B, kf = samp["valid"].shape[0], samp["k_fg"]
    K = rh.num_classes
    boxes = samp["boxes"][:, :kf]
    n = B * kf
    if pooled is None:
        pooled = pool_roi_features(rh, features, samp)[1]
    cube_features = pooled.flatten(1)
    raw, layout = rh.cube_head.forward_fused(cube_features)
    assert rh.use_confidence > 0 and rh.dims_priors_func == "exp"
    priors = rh.priors_dims_per_cat.detach()[0, :, 0, :].contiguous() if rh.dims_priors_enabled else None
    L, u_sel, dec, buf, validf = ops.cube_head_loss(raw, layout, K, samp["classes"], samp["valid"], samp["gt_idx"], kf,
                                                    gt.boxes3D, gt.poses, priors, meta, boxes.reshape(n, 4),
                                                    allocentric=rh.allocentric_pose, chamfer_pose=rh.chamfer_pose,
                                                    use_conf=True, joint=rh.loss_w_joint > 0, z_cfg=rh.z_cfg())
    red, stats = ops.cube_reduce(L, u_sel, buf, dec, validf, inverse_z=bool(rh.inverse_z_weight))
    p = "Cube/"
    w3 = rh.loss_w_3d
    # red = [dims, xy, z, pose, joint, uncert]: all six weighted by ONE vector multiply; the dict entries are views of it (their
    # gradients come back through one stack instead of a zero-fill + copy + add per term)
    wkey = (float(rh.loss_w_dims * w3), float(rh.loss_w_xy * w3), float(rh.loss_w_z * w3), float(rh.loss_w_pose * w3),
            float(rh.loss_w_joint * w3), float(rh.use_confidence), str(red.device))
    wv = _WVEC.get(wkey)
    if wv is None:
        wv = _WVEC[wkey] = torch.tensor(wkey[:6], dtype=torch.float32, device=red.device)
    scaled = (red * wv).unbind(0)
    losses = {p + "uncert": scaled[5], p + "loss_xy": scaled[1], p + "loss_z": scaled[2], p + "loss_pose": scaled[3]}
    if rh.loss_w_dims > 0:
        losses[p + "loss_dims"] = scaled[0]
    if rh.loss_w_joint > 0:
        losses[p + "loss_joint"] = scaled[4]
    storage = get_event_storage()
    storage.put_scalar(p + "z_error", stats[0], smoothing_hint=False)
    storage.put_scalar(p + "dims_error", stats[1], smoothing_hint=False)
    storage.put_scalar(p + "xy_error", stats[2], smoothing_hint=False)
    storage.put_scalar(p + "conf", stats[3], smoothing_hint=False)
    return losses


def forward_train(model, image_sizes, features, head_outputs, gt: GTBatch, meta):
    """RCNN3D.forward in training mode (rcnn3d.py:50-89) on the static-shape path.  gt: GTBatch; meta: camera_meta()."""
    rpn, rh = model.proposal_generator, model.roi_heads
    dev = features[rpn.in_features[0]].device
    feats = [features[f] for f in rpn.in_features]
    grid_sizes = [(f.shape[1], f.shape[2]) for f in feats]
    anchors_lv = rpn.anchor_generator(grid_sizes, dev)
    anchors = torch.cat([a.tensor for a in anchors_lv])
    logits, deltas, padded, sizes = rpn_tensors(rpn, feats, head_outputs)
    with torch.no_grad():
        labels, midx, _ = rpn_label_and_sample(rpn, anchors, gt)
    losses = rpn_losses(rpn, anchors, logits, deltas, labels, midx, gt)
    pboxes, pscores = rpn_proposals_padded(rpn, anchors, head_outputs[0] if padded is None else None, deltas, image_sizes,
                                           padded=padded, sizes=sizes)
    samp = roi_label_and_sample(rh, pboxes, pscores, gt)
    box_pooled, cube_pooled = pool_roi_features(rh, features, samp)
    lb, pred_boxes = box_head_losses(rh, features, samp, gt, pooled=box_pooled)
    losses.update(lb)
    if rh.loss_w_3d > 0:
        losses.update(cube_head_losses(rh, features, samp, pred_boxes, gt, meta, pooled=cube_pooled))
    return losses


_WEAK_FUSED_LOSSES = {"dims", "pose_alignment", "pose_ground", "iou", "z", "z_pseudo_gt_patch", "z_pseudo_gt_center"}
_WEAK_PRIOR_NAN = {}


def weak_fusable(rh, kf, dev, masks=None):
    """whether ops.weak_cube_loss covers this head's configuration: the loss set, uncertainty weighting on, 'exp' dimension
    priors, at most ops.WEAK_MAX_SLOTS foreground slots per image, no test hook installed (the hooks are CPU restatements of
    single kernels for the tensor composition); CR_WEAK_FUSED=0 selects the composition for comparisons."""
    import os
    if os.environ.get("CR_WEAK_FUSED", "1") == "0" or dev.type != "cuda":
        return False
    hooks = (rh._median_fn, rh._plane_cls, rh._ransac_triples, rh._hull_fn, rh._focal_fn)
    return (set(rh.loss_functions) <= _WEAK_FUSED_LOSSES and rh.use_confidence > 0 and kf <= ops.WEAK_MAX_SLOTS
            and (not rh.dims_priors_enabled or rh.dims_priors_func == "exp") and all(h is None for h in hooks))


def _prior_std_has_nan(rh):
    """dim_hinge_loss drops the three dims terms when a prior has a NaN standard deviation (roi_heads.py:1236-1237 looks at the
    selected priors; here: at the whole table, read back once per version of the parameter)"""
    p = rh.priors_dims_per_cat
    key = (id(p), p._version)
    if key not in _WEAK_PRIOR_NAN:
        _WEAK_PRIOR_NAN.clear()
        _WEAK_PRIOR_NAN[key] = bool(torch.isnan(p.detach()[0, :, 1, :]).any())
    return _WEAK_PRIOR_NAN[key]


def weak_cube_losses_fused(rh, samp, gt: GTBatch, cube_pooled, Ks, image_sizes, im_scales_ratio, ground_maps, depth_maps, generator=None,
                           raw_layout=None, normals=None):
    """ROIHeads3DScore._forward_cube in training (roi_heads.py:1366-1760) on the k_fg foreground slots of every image: the
    ground normals of the batch (one RANSAC launch), then ops.weak_cube_loss (select, terms, window medians, reduce).  Empty
    slots are excluded by their validity flag.  When every image holds exactly one foreground RoI the reference drops
    Cube/loss_pose from the dictionary; here it is reported as 0 (same total).  raw_layout = (raw, layout): the fused predictor
    output if the caller already has it; normals (B,3): ground normals computed elsewhere (both: tools and tests)."""
    from .roi_heads import weak_losses as W
    B, kf = samp["valid"].shape[0], samp["k_fg"]
    K = rh.num_classes
    n = B * kf
    lf = set(rh.loss_functions)
    raw, layout = raw_layout if raw_layout is not None else rh.cube_head.forward_fused(cube_pooled.flatten(1))
    dev = raw.device
    meta = camera_meta(rh, Ks, im_scales_ratio, image_sizes, dev)
    table = W.weak_table(Ks, im_scales_ratio, image_sizes, ground_maps, depth_maps).to(dev, non_blocking=True)
    prior_mean = prior_std = None
    if rh.dims_priors_enabled:
        pr = rh.priors_dims_per_cat.detach()[0]
        prior_mean, prior_std = pr[:, 0, :].contiguous(), pr[:, 1, :].contiguous()
    terms = 0
    if "iou" in lf:
        terms |= 1
    if "pose_alignment" in lf:
        terms |= 2
    if "pose_ground" in lf:
        terms |= 4
    if "z" in lf:
        terms |= 8
    pgz = 1 if "z_pseudo_gt_patch" in lf else (2 if "z_pseudo_gt_center" in lf else 0)
    if pgz:
        terms |= 16
    if "dims" in lf and not (rh.dims_priors_enabled and _prior_std_has_nan(rh)):
        terms |= 32 | 64 | 128
    if (terms & 4) and normals is None:
        # the reference back-projects image i with the intrinsics of the i-th foreground RoI's image (:1612 passes the per-box K)
        with torch.no_grad():
            cls = samp["classes"][:, :kf]
            vf = (samp["valid"][:, :kf] & (cls >= 0) & (cls < K)).reshape(-1)
            rank = torch.cumsum(vf, 0)
            hit = (rank.view(1, n) == torch.arange(1, B + 1, device=dev).view(B, 1)) & vf.view(1, n)
            first = torch.argmax(hit.to(torch.uint8), dim=1)
            own = torch.arange(B, device=dev)
            kimg = torch.where(hit.any(1), torch.div(first, kf, rounding_mode="floor"), own)
            normals = W.ground_normals_batched(ground_maps, depth_maps, table, kimg, generator=generator)
    w_log = (rh.loss_w_iou, rh.loss_w_pose, 0.0, rh.loss_w_z, rh.loss_w_z, rh.loss_w_dims, rh.loss_w_dims, rh.loss_w_dims,
             rh.loss_w_normal_vec)
    red, stats, _, _, _ = ops.weak_cube_loss(
        raw, layout, K, samp["classes"], samp["valid"], samp["gt_idx"], kf, gt.boxes, gt.boxes3D, gt.poses, prior_mean, prior_std,
        meta, table, normals, samp["boxes"][:, :kf].reshape(n, 4), depth_maps.tensor if pgz else None, terms, pgz, w_log,
        allocentric=rh.allocentric_pose, z_cfg=rh.z_cfg())
    w3 = rh.loss_w_3d
    wkey = ("weak", float(rh.loss_w_iou * w3), float(rh.loss_w_pose * w3), float(rh.loss_w_normal_vec * w3), float(rh.loss_w_z * w3),
            float(rh.loss_w_z * w3), float(rh.loss_w_dims * w3), float(rh.loss_w_dims * w3), float(rh.loss_w_dims * w3),
            float(rh.use_confidence), str(dev))
    wv = _WVEC.get(wkey)
    if wv is None:
        wv = _WVEC[wkey] = torch.tensor(wkey[1:10], dtype=torch.float32, device=dev)
    scaled = (red * wv).unbind(0)
    p = "Cube/"
    losses = {p + "uncert": scaled[8]}
    for k, name in enumerate(("loss_iou", "loss_pose", "loss_normal_vec", "loss_z", "loss_pseudo_gt_z", "loss_dims_w", "loss_dims_h",
                              "loss_dims_l")):
        if terms & (1 << k):
            losses[p + name] = scaled[k]
    storage = get_event_storage()
    for k, name in enumerate(("z_error", "dims_error", "xy_error", "z_close", "2D IoU", "conf")):
        storage.put_scalar(p + name, stats[k], smoothing_hint=False)
    storage.put_scalar(p + "total_3D_loss", stats[6] * w3, smoothing_hint=False)
    return losses


def forward_train_weak(model, image_sizes, features, head_outputs, gt: GTBatch, Ks, im_scales_ratio, ground_maps, depth_maps,
                       masks=None, mask_keys=None):
    """RCNN3D_combined_features.forward in training mode (rcnn3d.py:362-414): RPN losses, proposals, RoI sampling and
    the box head exactly as `forward_train` (fused, static shapes, no host sync).  The weak cube losses of
    ROIHeads3DScore run in the fused kernels of ops.weak_cube_loss on the (B, k_fg) foreground slots when the configured loss
    set is one they cover (weak_fusable) -- no host sync in the step; otherwise (segmentation / depth losses on object masks,
    pose_ground2) as the tensor composition of weak_losses.py on the COMPACTED foreground RoIs, whose number per image the host
    then has to wait for."""
    rpn, rh = model.proposal_generator, model.roi_heads
    dev = features[rpn.in_features[0]].device
    feats = [features[f] for f in rpn.in_features]
    grid_sizes = [(f.shape[1], f.shape[2]) for f in feats]
    anchors_lv = rpn.anchor_generator(grid_sizes, dev)
    anchors = torch.cat([a.tensor for a in anchors_lv])
    logits, deltas, padded, sizes = rpn_tensors(rpn, feats, head_outputs)
    with torch.no_grad():
        labels, midx, _ = rpn_label_and_sample(rpn, anchors, gt)
    losses = rpn_losses(rpn, anchors, logits, deltas, labels, midx, gt)
    pboxes, pscores = rpn_proposals_padded(rpn, anchors, head_outputs[0] if padded is None else None, deltas, image_sizes,
                                           padded=padded, sizes=sizes)
    samp = roi_label_and_sample(rh, pboxes, pscores, gt)
    box_pooled, cube_pooled = pool_roi_features(rh, features, samp)
    lb, _ = box_head_losses(rh, features, samp, gt, pooled=box_pooled)
    losses.update(lb)
    if rh.loss_w_3d <= 0:
        return losses
    B, kf = samp["valid"].shape[0], samp["k_fg"]
    if weak_fusable(rh, kf, dev, masks):
        losses.update(weak_cube_losses_fused(rh, samp, gt, cube_pooled, Ks, image_sizes, im_scales_ratio, ground_maps, depth_maps))
        return losses
    cls = samp["classes"][:, :kf]
    fg = samp["valid"][:, :kf] & (cls >= 0) & (cls < rh.num_classes)
    counts = fg.sum(1).tolist()                                   # host sync
    if sum(counts) == 0:
        return losses
    sel = torch.nonzero(fg.reshape(-1)).squeeze(1)                # image-major order
    img = torch.div(sel, kf, rounding_mode="floor")
    pick = lambda t: t[:, :kf].reshape(B * kf, *t.shape[2:])[sel]
    gidx = pick(samp["gt_idx"])
    lc, _, _ = rh.weak_losses_flat(cube_pooled.flatten(1)[sel], pick(samp["classes"]), pick(samp["boxes"]), gt.boxes[img, gidx],
                                   gt.boxes3D[img, gidx], gt.poses[img, gidx], counts, Ks, image_sizes, im_scales_ratio,
                                   ground_maps, depth_maps, masks, mask_keys)
    losses.update(lc)
    return losses
