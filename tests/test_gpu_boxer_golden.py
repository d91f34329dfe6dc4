"""GPU: the ground-truth-box branches of ROIHeads_Boxer -- MABO diagnostics (`output_recall_scores`), the two pseudo-ground-
truth modes of training and the AP packing on GT boxes -- against the outputs of the REFERENCE's own
ROIHeads_Boxer._forward_cube (cubercnn/modeling/roi_heads/roi_heads.py:304-660; tests/golden/make_golden_boxer.py ran it in the
build container with the third-party pieces stood in: see the fixtures' `notes`).  The proposals the reference sampled are part
of the fixture (`predict_cubes` is replayed), everything downstream runs on the HIP kernels through the C ABI: projection,
IoU2D against the projected ground-truth cube, size prior, corner chamfer with the mask rectangles of cr_mask_rects, exact
IoU3D (cr_box3d_overlap), point-in-box counts, the raster counts of the mask scores, then the host-side ranking tables."""
import importlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
d2 = importlib.import_module("3dod_amd.d2lite")
syn = importlib.import_module("3dod_amd.synthetic")
modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
spaces = importlib.import_module("3dod_amd.ProposalNetwork.utils.spaces")
conv = importlib.import_module("3dod_amd.ProposalNetwork.utils.conversions")


def _head(g):
    cfg_file = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "BoxNet.yaml")
    cfg = syn.make_cfg(cfg_file, ["MODEL.DEVICE", "cuda:0", "VIS_PERIOD", 0, "log", False])
    shapes = {f"p{l}": d2.ShapeSpec(channels=256, stride=2 ** l) for l in range(2, 7)}
    rh = modeling.build_roi_heads(cfg, shapes).to(DEV)
    rh.priors_dims_per_cat.data = torch.tensor(g["priors"]).to(DEV)
    cubes = spaces.Cubes(torch.tensor(g["cubes"]).to(DEV))
    stats = (None if g["stats_image"].ndim == 0 else torch.tensor(g["stats_image"]),
             None if g["stats_ranges"].ndim == 0 else g["stats_ranges"])

    def predict_cubes(gt_boxes, priors, depth, im_shape, K, fn, normal, gt_3d=None, generator=None):
        return cubes, conv.cubes_to_box(cubes, K, im_shape), stats[0], stats[1]       # the reference's own draws, replayed
    rh.predict_cubes = predict_cubes
    inst = d2.Instances(tuple(g["depth"].shape))
    inst.gt_boxes = d2.Boxes(torch.tensor(g["gt_boxes"]).to(DEV))
    inst.gt_classes = torch.tensor(g["gt_classes"]).to(DEV)
    inst.gt_boxes3D = torch.tensor(g["gt_boxes3D"]).to(DEV)
    inst.gt_poses = torch.tensor(g["gt_poses"]).to(DEV)
    return rh, inst


def _run(g, training, ex):
    rh, inst = _head(g)
    rh.train(training)
    np.random.seed(int(g["seed"]))                     # the random ranking score comes from numpy's global stream
    depth, ground = torch.tensor(g["depth"]).to(DEV)[None], torch.tensor(g["ground"]).to(DEV)[None]
    masks = torch.tensor(g["masks"]).to(DEV)
    return rh._gt_modes_one_image(tuple(g["depth"].shape), inst, depth, ground, torch.tensor(g["K"]), 1.0, masks, None, "propose",
                                  dict(ex, use_pred_boxes=False))


def _close(a, b, rtol=1e-4, atol=1e-5):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


def _mostly(a, b, frac, rtol=1e-4, atol=1e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    ok = np.abs(a - b) <= atol + rtol * np.abs(b)
    assert ok.mean() >= frac, (ok.mean(), a.shape)


def test_mabo_branch_matches_the_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "boxer_mabo.npz"), allow_pickle=False)
    out = _run(g, False, {"output_recall_scores": True})
    (p_info, s_iou, s_seg, s_dim, s_comb, s_rand, s_pc, empty, stats_image, stats_off, s_segm, s_cor, comb) = out
    # the chosen cube of every object: the argmax of IoU2D x dims x corners is the reference's
    _close(p_info.pred_cubes.tensor.cpu().numpy(), g["out_cubes"], rtol=1e-6, atol=1e-6)
    _close(p_info.pred_cubes.scores.cpu().numpy(), g["out_scores"])
    _close(p_info.pred_boxes.tensor.cpu().numpy(), g["out_pred_boxes"], atol=2e-2)
    # ranking tables: entry k = best IoU3D among the k+1 top-scoring proposals.  Two proposals whose scores differ in the
    # last bit may swap ranks, which moves single entries; the tables must agree almost everywhere, at rank 1 and at the end
    for name, got in (("score_IoU2D", s_iou), ("score_seg", s_seg), ("score_dim", s_dim), ("score_combined", s_comb),
                      ("score_random", s_rand), ("score_point_c", s_pc), ("score_seg_mod", s_segm), ("score_corner", s_cor)):
        ref = g[name]
        assert got.shape == ref.shape == (len(g["gt_boxes"]), 1000), name
        _mostly(got, ref, 0.97)
        _close(got[:, -1], ref[:, -1])                 # the best IoU3D over all proposals
        assert (np.diff(got, axis=1) >= 0).all()
    _close(s_comb[:, 0], g["score_combined"][:, 0])    # MABO at rank 1 of the method's own score
    _close(s_rand, g["score_random"])                  # the same numpy stream ranks the same proposals
    _mostly(comb, g["combinations"], 0.9)
    _close(stats_off, g["stats_off"], rtol=1e-3, atol=1e-4)
    assert abs(empty - float(g["stat_empty_boxes"])) <= 1.0
    _close(np.asarray(stats_image), g["stats_image"], rtol=1e-4, atol=1e-5)


def test_pseudo_gt_modes_match_the_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "boxer_pseudo.npz"), allow_pickle=False)
    (r,) = _run(g, True, {"pseudo_gt": "pseudo"})
    for f in ("scores", "pred_bbox3D", "pred_center_cam", "pred_dimensions", "pred_pose", "pred_center_2D"):
        _close(r.get(f).cpu().numpy(), g["out_" + f], atol=2e-3 if f == "pred_center_2D" else 1e-5)
    assert (r.pred_classes.cpu().numpy() == g["out_pred_classes"]).all()
    _close(r.pred_boxes.tensor.cpu().numpy(), g["out_pred_boxes"], atol=2e-2)
    g = np.load(os.path.join(golden_dir, "boxer_learn.npz"), allow_pickle=False)
    cubes = _run(g, True, {"pseudo_gt": "learn"})
    _close(cubes.tensor.cpu().numpy(), g["out_cubes"], rtol=1e-6, atol=1e-6)
    _close(cubes.scores.cpu().numpy(), g["out_scores"], atol=2e-5)
    with pytest.raises(ValueError, match="pseudo_gt"):
        _run(g, True, {})


def test_ap_packing_on_gt_boxes_matches_the_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "boxer_ap_gt.npz"), allow_pickle=False)
    (r,) = _run(g, False, {})
    for f in ("scores", "pred_bbox3D", "pred_center_cam", "pred_dimensions", "pred_pose", "pred_center_2D"):
        _close(r.get(f).cpu().numpy(), g["out_" + f], atol=1e-5)
    _close(r.pred_boxes.tensor.cpu().numpy(), g["out_pred_boxes"], atol=1e-5)       # the GT boxes themselves (:652)


def test_boxnet_model_routes_experiment_types():
    """BoxNet.forward (rcnn3d.py:678-713) reaches the three GT-box modes through the model: MABO tuple in eval mode, pseudo
    ground truth in training mode, a list of IoU3D tables for a list of proposal functions"""
    cfg_file = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "BoxNet.yaml")
    cfg = syn.make_cfg(cfg_file, ["MODEL.DEVICE", "cuda:0", "VIS_PERIOD", 0, "log", False, "MODEL.ROI_CUBE_HEAD.NUMBER_OF_PROPOSALS", 200])
    torch.manual_seed(0)
    model = modeling.build_model(cfg).eval()
    b = syn.make_batch(1, 5, size=256, min_obj=3, max_obj=3)
    g = torch.Generator().manual_seed(2)
    b[0]["depth_map"] = torch.rand(256, 256, generator=g) * 3 + 1
    b[0]["ground_map"] = (torch.arange(256)[:, None] > 150).expand(256, 256).to(torch.uint8)
    m = torch.zeros(3, 256, 256, dtype=torch.bool)
    for j, bb in enumerate(b[0]["instances"].gt_boxes.tensor.round().long().clamp(0, 255)):
        m[j, bb[1]:bb[3] + 1, bb[0]:bb[2] + 1] = True
    b[0]["masks"] = m
    out = model(b, experiment_type={"use_pred_boxes": False, "output_recall_scores": True})
    assert len(out) == 13 and out[1].shape == (3, 200) and out[12].shape == (3, 26)
    tables = model(b, experiment_type={"use_pred_boxes": False, "output_recall_scores": True}, proposal_function=["propose", "random"])
    assert tuple(tables.shape) == (3, 2, 200) and float(tables.min()) >= 0 and float(tables.max()) <= 1.0 + 1e-6
    model.train()
    (r,) = model(b, experiment_type={"pseudo_gt": "pseudo"})
    assert len(r) == 3 and r.pred_bbox3D.shape == (3, 8, 3)
    learn = model(b, experiment_type={"pseudo_gt": "learn"})
    assert tuple(learn.tensor.shape) == (3, 200, 15) and tuple(learn.scores.shape) == (3, 200)
