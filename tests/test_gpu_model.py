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
