"""GPU: the test-time filter of the box head without a host round trip (ops.det_select = cr_det_scores, cr_topk, cr_det_gather,
cr_nms_grouped_cls, cr_det_pick) against fast_rcnn_inference_single_image (fast_rcnn.py:57-116 restated; itself pinned to the
reference's function by the G7 golden, tests/test_gpu_dense_golden.py): same detections in the same order, per image."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
ops = importlib.import_module("3dod_amd.hipops")
d2 = importlib.import_module("3dod_amd.d2lite")
fr = importlib.import_module("3dod_amd.cubercnn.modeling.roi_heads.fast_rcnn")
W4, CLAMP = (10.0, 10.0, 5.0, 5.0), 4.135166556742356


def _case(seed, B=3, P=300, K=20, agnostic=False):
    g = torch.Generator().manual_seed(seed)
    sizes = [(384, 512), (512, 448), (512, 512)][:B]
    logits = torch.randn(B * P, K + 1, generator=g)
    hot = torch.rand(B * P, generator=g) < 0.5                         # half of the rows: one or two confident classes
    c1 = torch.randint(0, K, (B * P,), generator=g)
    logits[hot, c1[hot]] += torch.rand(int(hot.sum()), generator=g) * 6 + 2
    c2 = torch.randint(0, K, (B * P,), generator=g)
    two = hot & (torch.rand(B * P, generator=g) < 0.3)
    logits[two, c2[two]] += 4.0
    ctr = torch.rand(B * P, 2, generator=g) * 400 + 40
    wh = torch.rand(B * P, 2, generator=g) * 120 + 16
    # clusters of overlapping proposals so that NMS has work to do
    ctr[::3] = ctr[1::3][:ctr[::3].shape[0]] + torch.randn(ctr[::3].shape, generator=g) * 4
    wh[::3] = wh[1::3][:wh[::3].shape[0]]
    props = torch.cat((ctr - wh / 2, ctr + wh / 2), 1)
    deltas = torch.randn(B * P, 4 if agnostic else 4 * K, generator=g) * 0.3
    obj = torch.randn(B * P, generator=g)
    obj[P - 17:P] = float("-inf")                                       # padding slots of image 0
    obj[3 * P - 40:] = float("-inf") if B >= 3 else obj[3 * P - 40:]
    logits[5, 3] = float("nan")                                         # a non-finite prediction is dropped
    deltas[9, 2] = float("inf")
    return dict(B=B, P=P, K=K, sizes=sizes, logits=logits.to(DEV), deltas=deltas.to(DEV), props=props.to(DEV), obj=obj.to(DEV))


def _reference(c, thresh, nms, topk):
    t = d2.Box2BoxTransform(weights=W4, scale_clamp=CLAMP)
    out = []
    P = c["P"]
    for b in range(c["B"]):
        sl = slice(b * P, (b + 1) * P)
        boxes = t.apply_deltas(c["deltas"][sl], c["props"][sl])
        probs = torch.softmax(c["logits"][sl], dim=-1)
        probs = torch.where(torch.isfinite(c["obj"][sl])[:, None], probs, torch.full_like(probs, float("nan")))
        inst, kept = fr.fast_rcnn_inference_single_image(boxes, probs, c["sizes"][b], thresh, nms, topk)
        # the function indexes the rows that survive its finiteness filter (fast_rcnn.py:75-78): back to proposal slots
        valid = torch.isfinite(boxes).all(1) & torch.isfinite(probs).all(1)
        out.append((inst, torch.nonzero(valid).squeeze(1)[kept]))
    return out


def _run(c, thresh, nms, topk, max_candidates=2048):
    hw = torch.tensor([[float(h), float(w)] for h, w in c["sizes"]], device=DEV)
    return ops.det_select(c["logits"], c["deltas"], c["props"], c["obj"], hw, c["B"], c["P"], c["K"], W4, CLAMP, thresh, nms, topk,
                          max_candidates=max_candidates)


@pytest.mark.parametrize("agnostic", [False, True])
def test_det_select_matches_the_per_image_filter(agnostic):
    c = _case(3, agnostic=agnostic)
    ref = _reference(c, 0.05, 0.5, 100)
    boxes, scores, cls, rows, full, cnt = _run(c, 0.05, 0.5, 100)
    cnt = cnt.cpu()
    assert int(cnt[:, 1].sum()) == 0
    for b, (inst, kept_rows) in enumerate(ref):
        n = int(cnt[b, 0])
        assert n == len(inst) and n > 20, (b, n, len(inst))
        assert torch.equal(cls[b, :n], inst.pred_classes)
        assert torch.equal(rows[b, :n], kept_rows)
        assert torch.allclose(scores[b, :n], inst.scores, atol=1e-6, rtol=1e-6)
        assert torch.allclose(boxes[b, :n], inst.pred_boxes.tensor, atol=1e-3, rtol=1e-5)
        assert torch.allclose(full[b, :n], inst.scores_full, atol=1e-6, rtol=1e-5)
        assert float(scores[b, n:].abs().max() if n < 100 else 0.0) == 0.0          # empty slots are zero
        assert bool((scores[b, :n - 1] >= scores[b, 1:n]).all())


def test_det_select_topk_cut_and_overflow_flag():
    c = _case(4)
    # a low threshold: thousands of candidates per image, the 30 best survivors are wanted
    ref = _reference(c, 0.01, 0.5, 30)
    boxes, scores, cls, rows, full, cnt = _run(c, 0.01, 0.5, 30)
    cnt = cnt.cpu()
    for b, (inst, kept_rows) in enumerate(ref):
        assert int(cnt[b, 0]) == len(inst) == 30 and int(cnt[b, 1]) == 0
        assert torch.equal(cls[b], inst.pred_classes) and torch.equal(rows[b], kept_rows)
        assert torch.allclose(scores[b], inst.scores, atol=1e-6)
    # 64 candidates are not enough to find 100 survivors: the images are flagged for the unbounded path
    _, _, _, _, _, cnt2 = _run(c, 0.01, 0.5, 100, max_candidates=64)
    cnt2 = cnt2.cpu()
    assert bool((cnt2[:, 1] == 1).all()) and bool((cnt2[:, 0] <= 64).all())
    # no candidate at all
    _, s3, _, _, _, cnt3 = _run(c, 1.1, 0.5, 100)
    assert int(cnt3.cpu().sum()) == 0 and float(s3.abs().max()) == 0.0


def test_model_inference_fused_equals_the_per_image_path(monkeypatch):
    """RCNN3D.inference through ROIHeads3D._infer_padded (one host wait) == the per-image path (CR_INFER_FUSED=0) on the same
    weights and images: same detections, 2D and 3D fields"""
    syn = importlib.import_module("3dod_amd.synthetic")
    modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
    cfg = syn.make_cfg(overrides=["MODEL.DEVICE", "cuda:0", "VIS_PERIOD", 0, "log", False])
    torch.manual_seed(1)
    model = modeling.build_model(cfg).eval()
    model.roi_heads.box_predictor.test_score_thresh = 0.0205           # random-init scores sit around 1 / 51
    batch = syn.make_batch(3, 41, with_gt=False)
    calls = []
    orig = ops.det_select
    monkeypatch.setattr(ops, "det_select", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    with torch.no_grad():
        fused = model(batch)
        assert calls, "inference did not take the fused path"
        monkeypatch.setenv("CR_INFER_FUSED", "0")
        plain = model(batch)
    assert sum(len(o["instances"]) for o in plain) > 50
    for a, b in zip(fused, plain):
        ia, ib = a["instances"], b["instances"]
        assert len(ia) == len(ib)
        assert torch.equal(ia.pred_classes, ib.pred_classes)
        assert torch.allclose(ia.pred_boxes.tensor, ib.pred_boxes.tensor, atol=1e-3)
        for f in ("scores", "scores_full", "pred_bbox3D", "pred_center_cam", "pred_center_2D", "pred_dimensions", "pred_pose"):
            assert torch.allclose(ia.get(f), ib.get(f), atol=1e-4, rtol=1e-4), f
