"""GPU: the per-image labelling steps under their reference names and signatures (RPNWithIgnore.label_and_sample_anchors,
rpn.py:41-110; ROIHeads3D.label_and_sample_proposals, roi_heads.py:2773-2840) are thin wrappers over the fused kernels of the
static-shape path.  The sampling is random (its distribution is tested in test_gpu_dense_train.py / test_dense_train.py
against the reference's own functions); here: the contract of the two methods -- shapes, the matching rules every draw must
satisfy, the copied ground-truth fields."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
d2 = importlib.import_module("3dod_amd.d2lite")
syn = importlib.import_module("3dod_amd.synthetic")
modeling = importlib.import_module("3dod_amd.cubercnn.modeling")


def _iou(a, b):
    lt, rb = torch.max(a[:, None, :2], b[None, :, :2]), torch.min(a[:, None, 2:], b[None, :, 2:])
    inter = (rb - lt).clamp(min=0).prod(-1)
    area = lambda x: (x[:, 2] - x[:, 0]) * (x[:, 3] - x[:, 1])
    return inter / (area(a)[:, None] + area(b)[None, :] - inter)


def test_label_and_sample_under_the_reference_names():
    cfg = syn.make_cfg(overrides=["MODEL.DEVICE", "cuda:0", "VIS_PERIOD", 0, "log", False])
    torch.manual_seed(0)
    model = modeling.build_model(cfg).train()
    pg, rh = model.proposal_generator, model.roi_heads
    batch = syn.make_batch(2, 31, min_obj=3, max_obj=6)
    gts = [b["instances"].to(DEV) for b in batch]
    grid = [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)]
    anchors = pg.anchor_generator(grid, DEV)
    with d2.EventStorage(0):
        labels, boxes = pg.label_and_sample_anchors(anchors, gts)
    A = sum(len(a) for a in anchors)
    allA = d2.Boxes.cat(anchors).tensor
    assert len(labels) == len(boxes) == 2
    for l, mb, g in zip(labels, boxes, gts):
        assert l.shape == (A,) and l.dtype == torch.int8 and mb.shape == (A, 4)
        assert set(l.unique().tolist()) <= {-1, 0, 1}
        npos, nneg = int((l == 1).sum()), int((l == 0).sum())
        # the best anchor of every object is forced positive after the sampling (rpn.py:75): up to G more than the budget
        budget = pg.batch_size_per_image
        assert 0 < npos <= int(budget * pg.positive_fraction) + len(g) and budget <= npos + nneg <= budget + len(g)
        iou = _iou(g.gt_boxes.tensor, allA)                                  # (G, A)
        best = iou.max(0)
        # a sampled positive is above the foreground threshold or the best anchor of some object; its matched box is that object's
        pos = torch.nonzero(l == 1).squeeze(1)
        forced = (iou == iou.max(1, keepdim=True).values).any(0)
        assert bool(((best.values[pos] >= pg.anchor_matcher.thresholds[2]) | forced[pos]).all())
        assert torch.equal(mb[pos], g.gt_boxes.tensor[best.indices[pos]])
        neg = torch.nonzero(l == 0).squeeze(1)
        assert bool((best.values[neg] < pg.anchor_matcher.thresholds[2]).all())
    # RoI heads: 1000 proposals per image around the objects
    g = torch.Generator().manual_seed(4)
    props = []
    for t in gts:
        n = 1000
        ctr = torch.rand(n, 2, generator=g) * 480 + 16
        wh = torch.rand(n, 2, generator=g) * 150 + 10
        pb = torch.cat((ctr - wh / 2, ctr + wh / 2), 1).clamp(0, 512)
        pb[:len(t)] = t.gt_boxes.tensor.cpu() + 2.0                          # some near-perfect proposals
        p = d2.Instances((512, 512))
        p.proposal_boxes = d2.Boxes(pb.to(DEV))
        p.objectness_logits = torch.randn(n, generator=g).to(DEV)
        props.append(p)
    with d2.EventStorage(0):
        out = rh.label_and_sample_proposals(props, gts)
    assert len(out) == 2
    for inst, t in zip(out, gts):
        n = len(inst)
        assert 0 < n <= rh.batch_size_per_image and inst.gt_classes.shape == (n,)
        fg = (inst.gt_classes >= 0) & (inst.gt_classes < rh.num_classes)
        assert 0 < int(fg.sum()) <= int(rh.batch_size_per_image * rh.positive_fraction)
        k = int(fg.sum())
        assert bool(fg[:k].all()) and not bool(fg[k:].any())                 # foreground rows first
        for name in ("gt_boxes", "gt_boxes3D", "gt_poses"):
            assert inst.has(name) and len(inst.get(name)) == n
        iou = _iou(inst.proposal_boxes.tensor, t.gt_boxes.tensor)
        best = iou.max(1)
        assert bool((best.values[fg] >= rh.proposal_matcher.thresholds[1]).all())
        assert torch.equal(inst.gt_classes[fg], t.gt_classes[best.indices[fg]])
        assert torch.equal(inst.gt_boxes.tensor[fg], t.gt_boxes.tensor[best.indices[fg]])
        assert torch.equal(inst.gt_boxes3D[fg], t.gt_boxes3D[best.indices[fg]])
