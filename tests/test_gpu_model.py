"""GPU: the full Cube R-CNN DLA34-FPN train step and inference run end to end on the HIP kernels."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def built():
    bt = importlib.import_module("bench_train")
    return bt.build(DEV, seed=0)


def test_train_steps_run_and_learn(built):
    cfg, model, opt, syn, solver = built
    d2 = importlib.import_module("3dod_amd.d2lite")
    step = solver.TrainStep(cfg, model, opt, world_size=1)
    batch = syn.make_batch(2, 7)
    p0 = opt.flat_p.clone()
    totals = []
    with d2.EventStorage(0):
        for _ in range(6):
            step(batch)
            rep = step.report()
            totals.append(rep["total_loss"])
    assert all(t == t and abs(t) < 1e4 for t in totals), totals
    assert rep["iterations_explode"] == 0, rep
    assert not torch.equal(p0, opt.flat_p)
    expected = {"BoxHead/loss_cls", "BoxHead/loss_box_reg", "Cube/loss_dims", "Cube/loss_xy", "Cube/loss_z",
                "Cube/loss_pose", "Cube/loss_joint", "Cube/uncert", "rpn/cls", "rpn/loc"}
    assert expected <= set(rep.keys()), rep.keys()
    assert totals[-1] < totals[0], totals           # same batch 6 times: the loss goes down


def test_inference_runs(built):
    cfg, model, opt, syn, solver = built
    model.eval()
    thr = model.roi_heads.box_predictor.test_score_thresh
    model.roi_heads.box_predictor.test_score_thresh = -1.0     # barely-trained weights: keep detections alive
    try:
        with torch.no_grad():
            out = model(syn.make_batch(2, 9, with_gt=False))
    finally:
        model.roi_heads.box_predictor.test_score_thresh = thr
        model.train()
    assert len(out) == 2
    for o in out:
        inst = o["instances"]
        n = len(inst)
        assert 0 < n <= cfg.TEST.DETECTIONS_PER_IMAGE
        assert inst.pred_bbox3D.shape == (n, 8, 3) and inst.pred_pose.shape == (n, 3, 3)
        assert inst.pred_dimensions.shape == (n, 3) and inst.pred_center_cam.shape == (n, 3)
        assert inst.scores_full.shape == (n, cfg.MODEL.ROI_HEADS.NUM_CLASSES)
        assert torch.isfinite(inst.pred_bbox3D).all()


# ---------------------------------------------------------------------------------------------
# parity of the dense path: HIP (bf16 storage, f32 accumulate) vs float32 oracles
# ---------------------------------------------------------------------------------------------
def _rel(a, b):
    a = a.float().cpu(); b = b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12)), float((a - b).abs().max() / (b.abs().max() + 1e-12))


def test_dla_trunk_matches_reference_golden(golden_dir):
    """same seed -> same weights as the reference's DLA (tests/test_dla_weights.py); features vs the
    reference's own forward (tests/golden/make_golden_dla.py).  bf16 activations through 39 conv+BN layers:
    relative L2 <= 3e-2 per level (tolerance of the bf16 storage format, not of the arithmetic)."""
    import os
    import numpy as np
    dla = importlib.import_module("3dod_amd.cubercnn.modeling.backbone.dla")
    fpn = importlib.import_module("3dod_amd.cubercnn.modeling.backbone.fpn")
    ops = importlib.import_module("3dod_amd.hipops")
    g = np.load(os.path.join(golden_dir, "dla34_trunk.npz"), allow_pickle=False)
    torch.manual_seed(int(g["seed"]))
    net = fpn.to_channels_last(dla.dla34(pretrained=False)).to(DEV).train()
    x = torch.tensor(g["x"])
    xh = torch.cat([x.permute(0, 2, 3, 1), torch.zeros(x.shape[0], x.shape[2], x.shape[3], 5)], 3)
    xh = xh.to(DEV).to(torch.bfloat16).contiguous()
    with torch.no_grad():
        b = net.base_layer(xh)
        l1 = net.level1(net.level0(b))
        l2 = net.level2(l1); l3 = net.level3(l2); l4 = net.level4(l3); l5 = net.level5(l4)
    for name, got in (("base", b), ("level1", l1), ("p2", l2), ("p3", l3), ("p4", l4), ("p5", l5)):
        l2e, mx = _rel(got.permute(0, 3, 1, 2), torch.tensor(g[name]))
        assert l2e < 3e-2, (name, l2e, mx)


def test_backbone_rpn_head_match_float32_oracle(built):
    """whole FPN backbone + RPN head: the product on HIP vs the SAME host modules executed by the float32
    torch CPU backend (oracle/cpu_backend.py) with the same weights."""
    import copy
    from oracle import cpu_backend
    cfg, model, opt, syn, solver = built
    batch = syn.make_batch(2, 21, with_gt=False)
    model.train()
    with torch.no_grad():
        images, x = model.preprocess_image(batch)
        feats = model.backbone(x)
        logits, deltas = model.proposal_generator.rpn_head([feats[f] for f in model.proposal_generator.in_features])
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    # float32 CPU execution of the same modules: separate module instances, `ops` swapped in this process, restored after
    modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
    saved = {n: importlib.import_module(n).ops for n in cpu_backend.PATCHED}
    try:
        cpu_backend.install()
        cfg_cpu = syn.make_cfg(overrides=["MODEL.DEVICE", "cpu", "VIS_PERIOD", 0, "log", False])
        ref = modeling.build_model(cfg_cpu)
        ref.load_state_dict(sd)
        ref.train()
        with torch.no_grad():
            images_r, xr = ref.preprocess_image(batch)
            feats_r = ref.backbone(xr)
            logits_r, deltas_r = ref.proposal_generator.rpn_head([feats_r[f] for f in ref.proposal_generator.in_features])
    finally:
        for n, o in saved.items():
            importlib.import_module(n).ops = o
    for k in feats:
        l2e, mx = _rel(feats[k], feats_r[k])
        assert l2e < 4e-2, (k, l2e, mx)
    for a, b in zip(logits, logits_r):
        assert _rel(a, b)[0] < 6e-2
    for a, b in zip(deltas, deltas_r):
        assert _rel(a, b)[0] < 6e-2
