"""GPU: the fused training-glue kernels (csrc/dense_train.hip: k_box_match, k_rpn_label, k_rpn_scatter, k_rpn_loss,
k_roi_label, k_roi_compact, k_box_loss), the inference filter and the 3D head's FC stack, through the C ABI, against the
REFERENCE'S OWN outputs (tests/golden/dense_train_g7.npz; generator tests/golden/make_golden_dense.py; shared checks
tests/g7_checks.py).  Integers exact, IoUs 1e-6, losses / gradients 1e-5."""
import importlib

import pytest
import torch

import g7_checks as C

pytestmark = pytest.mark.gpu
ops = importlib.import_module("3dod_amd.hipops")
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def G():
    return C.load()


@pytest.mark.parametrize("tag", ["rpn", "rpnh"])
def test_rpn_labels_sampling_and_losses_match_reference(G, tag):
    C.check_rpn(ops, DEV, G, tag)


def test_roi_sampling_and_box_losses_match_reference(G):
    C.check_roi_and_box_loss(ops, DEV, G)


def test_inference_filter_matches_reference(G):
    fr = importlib.import_module("3dod_amd.cubercnn.modeling.roi_heads.fast_rcnn")
    C.check_inference_filter(fr, DEV, G)


def test_cube_head_forward_matches_reference(G, precision):
    """fp32 (the reference's precision): 1e-5; bf16 fast mode: 2e-2"""
    ch = importlib.import_module("3dod_amd.cubercnn.modeling.roi_heads.cube_head")
    syn = importlib.import_module("3dod_amd.synthetic")
    d2 = importlib.import_module("3dod_amd.d2lite")
    C.check_cube_head(ch, syn.make_cfg, d2, DEV, G, tol=1e-5 if precision != "bf16" else 3e-2)


def test_cube_head_with_per_predictor_trunks_matches_reference(golden_dir):
    """SHARED_FC = False on the GPU kernels (five FC chains, outputs side by side in the fused layout) against the reference's
    own CubeHead; and one train step of the model under that setting"""
    import os
    ch = importlib.import_module("3dod_amd.cubercnn.modeling.roi_heads.cube_head")
    syn = importlib.import_module("3dod_amd.synthetic")
    d2 = importlib.import_module("3dod_amd.d2lite")
    C.check_cube_head(ch, syn.make_cfg, d2, DEV, C.load(os.path.join(golden_dir, "cubehead_nonshared.npz")), shared_fc=False)
    bt = importlib.import_module("bench_train")
    cfg, model, opt, syn2, solver = bt.build(DEV, extra=["MODEL.ROI_CUBE_HEAD.SHARED_FC", False])
    assert hasattr(model.roi_heads.cube_head, "feature_generator_Z") and not hasattr(model.roi_heads.cube_head, "feature_generator")
    step = solver.TrainStep(cfg, model, opt, world_size=1)
    with d2.EventStorage(0):
        step(syn2.make_batch(2, 3))
        rep = step.report()
    assert rep["iterations_explode"] == 0 and rep["total_loss"] == rep["total_loss"]


@pytest.mark.parametrize("z_type,suffix", [("direct", ""), ("sigmoid", "_zsigmoid"), ("log", "_zlog"), ("direct", "_bins3_direct"), ("clusters", "_bins3_clusters")])
def test_cube_decode_infer_kernel_matches_reference_eval_golden(golden_dir, z_type, suffix):
    """cr_cube_decode_infer (the fused inference decode of the 3D head, roi_heads.py:2353-2436,2682-2735) against
    tests/golden/cubehead_eval.npz = the reference's own ROIHeads3D._forward_cube in eval mode: corners, centres,
    dimensions, pose and merged scores within 1e-4 relative (north_star's corner tolerance)."""
    import os
    import numpy as np
    util = importlib.import_module("3dod_amd.cubercnn.util.math_util")
    g = np.load(os.path.join(golden_dir, "cubehead_eval%s.npz" % suffix), allow_pickle=False)      # Z_TYPE variants: roi_heads.py:2404-2410
    T = lambda k: torch.tensor(g[k]).to(DEV)
    n, K = g["in_deltas"].shape[0], g["in_deltas"].shape[1]
    bins = g["priors_z_scales"].shape[1] if "priors_z_scales" in g.files else 1        # CLUSTER_BINS: z is (n, bins, K, 1)
    ld = ((12 + bins) * K + 15) // 16 * 16
    raw = torch.zeros((n, ld), device=DEV)
    raw[:, 0:2 * K] = T("in_deltas").reshape(n, -1)
    raw[:, 2 * K:5 * K] = T("in_dims").reshape(n, -1)
    raw[:, 5 * K:11 * K] = T("in_pose6").reshape(n, -1)
    raw[:, 11 * K:(11 + bins) * K] = T("in_z").reshape(n, -1)
    raw[:, (11 + bins) * K:(12 + bins) * K] = T("in_uncert")
    layout = (0, 2 * K, 5 * K, 11 * K, (11 + bins) * K)
    zc = ops.z_config(z_type, bins, T("priors_z_scales") if bins > 1 else None, T("priors_z_stats") if bins > 1 else None)
    n_per = g["n_per"].tolist()
    img = torch.repeat_interleave(torch.arange(len(n_per)), torch.tensor(n_per)).to(DEV)
    rows = []
    for k, r in zip(g["Ks"], g["ratios"]):
        r = float(r)
        v2r = util.compute_virtual_scale_from_focal_spaces(float(k[1, 1]), 512.0 * r, 512.0, 512.0)
        rows.append([float(k[0, 0]) / r, float(k[1, 1]) / r, float(k[0, 2]) / r, float(k[1, 2]) / r, float(v2r), r])
    meta6 = torch.tensor(rows, dtype=torch.float32, device=DEV)
    priors = T("priors")[0, :, 0, :].contiguous()
    # (in eval mode the 2D detections are the source boxes of the deltas and of the cluster bin, roi_heads.py:2257-2262)
    o = ops.cube_decode_infer(raw, layout, K, T("classes"), img, T("pred_boxes"), meta6, priors, allocentric=True,
                              z_cfg=zc).cpu().numpy()
    chk = lambda got, key: np.testing.assert_allclose(got, g[key], rtol=1e-4, atol=1e-5, err_msg=key)
    chk(o[:, 18:42].reshape(n, 8, 3), "out_pred_bbox3D")
    chk(o[:, 0:3], "out_pred_center_cam")
    chk(o[:, 6:8], "out_pred_center_2D")
    chk(o[:, 3:6], "out_pred_dimensions")
    chk(o[:, 9:18].reshape(n, 3, 3), "out_pred_pose")
    chk(np.sqrt(g["scores_2d"] * o[:, 8]), "out_scores")


@pytest.mark.parametrize("z_type,suffix", [("direct", ""), ("sigmoid", "_zsigmoid"), ("log", "_zlog"), ("direct", "_bins3_direct"), ("clusters", "_bins3_clusters")])
def test_dense_cube_head_loss_matches_reference_train_golden(golden_dir, z_type, suffix):
    """the static-shape training form of the 3D head -- cr_cube_select (class gather, 6D -> R, uncertainty clip, Z_TYPE decode),
    cr_cube_loss_fwd / _bwd, cr_cube_reduce, cr_cube_select_bwd through ops.cube_head_loss / cube_reduce -- on the (B, kf)
    slots of the golden's three images against the reference's own ROIHeads3D._forward_cube: the six reduced losses and the
    gradients w.r.t. every head output (tests/golden/cubehead_train*.npz)"""
    import os
    import numpy as np
    util = importlib.import_module("3dod_amd.cubercnn.util.math_util")
    g = np.load(os.path.join(golden_dir, "cubehead_train%s.npz" % suffix), allow_pickle=False)
    T = lambda k: torch.tensor(g[k]).to(DEV)
    n_per = g["n_per"].tolist()
    B, kf, K = len(n_per), max(n_per) + 2, g["in_deltas"].shape[1]
    bins = g["priors_z_scales"].shape[1] if "priors_z_scales" in g.files else 1
    n = B * kf
    ld = ((12 + bins) * K + 15) // 16 * 16
    slot = torch.cat([torch.arange(c) + b * kf for b, c in enumerate(n_per)]).to(DEV)          # golden row -> dense slot
    raw = torch.zeros((n, ld), device=DEV)
    src = torch.zeros((sum(n_per), ld), device=DEV)
    src[:, 0:2 * K] = T("in_deltas").reshape(-1, 2 * K)
    src[:, 2 * K:5 * K] = T("in_dims").reshape(-1, 3 * K)
    src[:, 5 * K:11 * K] = T("in_pose6").reshape(-1, 6 * K)
    src[:, 11 * K:(11 + bins) * K] = T("in_z").reshape(-1, bins * K)
    src[:, (11 + bins) * K:(12 + bins) * K] = T("in_uncert")
    raw[slot] = src
    raw.requires_grad_(True)
    layout = (0, 2 * K, 5 * K, 11 * K, (11 + bins) * K)
    zc = ops.z_config(z_type, bins, T("priors_z_scales") if bins > 1 else None, T("priors_z_stats") if bins > 1 else None)
    S = kf + 3
    cls = torch.full((B, S), K, dtype=torch.int64, device=DEV)
    valid = torch.zeros((B, S), dtype=torch.bool, device=DEV)
    gt_idx = torch.zeros((B, S), dtype=torch.int64, device=DEV)
    G = max(n_per)
    gt3d = torch.zeros((B, G, 9), device=DEV)
    gtpose = torch.eye(3, device=DEV).expand(B, G, 3, 3).clone()
    boxes = torch.zeros((B, kf, 4), device=DEV)
    boxes[..., 2:] = 10.0
    off = 0
    for b, c in enumerate(n_per):                                            # every RoI gets its own ground-truth row
        cls[b, :c] = T("gt_classes")[off:off + c]
        valid[b, :c] = True
        gt_idx[b, :c] = torch.arange(c, device=DEV)
        gt3d[b, :c] = T("gt_boxes3D")[off:off + c]
        gtpose[b, :c] = T("gt_poses")[off:off + c]
        boxes[b, :c] = T("proposal_boxes")[off:off + c]
        off += c
    rows = []
    for k, r in zip(g["Ks"], g["ratios"]):
        r = float(r)
        v2r = util.compute_virtual_scale_from_focal_spaces(float(k[1, 1]), 512.0 * r, 512.0, 512.0)
        rows.append([float(k[0, 0]) / r, float(k[1, 1]) / r, float(k[0, 2]) / r, float(k[1, 2]) / r, float(v2r)])
    meta = torch.tensor(rows, dtype=torch.float32, device=DEV)
    priors = T("priors")[0, :, 0, :].contiguous()
    L, u_sel, dec, buf, validf = ops.cube_head_loss(raw, layout, K, cls, valid, gt_idx, kf, gt3d, gtpose, priors, meta,
                                                    boxes.reshape(n, 4), allocentric=True, chamfer_pose=True, use_conf=True, joint=True,
                                                    z_cfg=zc)
    red, _ = ops.cube_reduce(L, u_sel, buf, dec, validf, inverse_z=False)
    # weights of make_golden_cubehead.py: dims 20, xy 1, z 1, pose 7, joint 1, uncertainty 1 (x loss_w_3d 1)
    w = torch.tensor([20.0, 1.0, 1.0, 7.0, 1.0, 1.0], device=DEV)
    names = ["loss_dims", "loss_xy", "loss_z", "loss_pose", "loss_joint", "uncert"]
    for i, nm in enumerate(names):
        ref = float(g["loss_Cube_" + nm])
        assert abs(float(red[i] * w[i]) - ref) <= 2e-5 * max(1.0, abs(ref)), (nm, float(red[i] * w[i]), ref)
    (red * w).sum().backward()
    gr = raw.grad[slot]
    np.testing.assert_allclose(gr[:, 0:2 * K].reshape(-1, K, 2).cpu().numpy(), g["grad_deltas"], rtol=5e-4, atol=5e-6)
    np.testing.assert_allclose(gr[:, 2 * K:5 * K].reshape(-1, K, 3).cpu().numpy(), g["grad_dims"], rtol=5e-4, atol=5e-6)
    np.testing.assert_allclose(gr[:, 5 * K:11 * K].reshape(-1, K, 6).cpu().numpy(), g["grad_pose6"], rtol=5e-4, atol=5e-6)
    np.testing.assert_allclose(gr[:, 11 * K:(11 + bins) * K].reshape(g["grad_z"].shape).cpu().numpy(), g["grad_z"], rtol=5e-4, atol=5e-6)
    np.testing.assert_allclose(gr[:, (11 + bins) * K:(12 + bins) * K].cpu().numpy(), g["grad_uncert"], rtol=5e-4, atol=5e-6)
    empty = torch.ones(n, dtype=torch.bool, device=DEV)
    empty[slot] = False
    assert float(raw.grad[empty].abs().max()) == 0.0


def test_model_trains_and_infers_with_cluster_bins_and_cluster_depth():
    """MODEL.ROI_CUBE_HEAD.CLUSTER_BINS = 3 with Z_TYPE 'clusters' end to end: the depth predictor has K * 3 outputs, train step
    and inference run on the fused kernels; the parameters the reference adds exist under its names"""
    bt = importlib.import_module("bench_train")
    d2 = importlib.import_module("3dod_amd.d2lite")
    cfg, model, opt, syn2, solver = bt.build(DEV, extra=["MODEL.ROI_CUBE_HEAD.CLUSTER_BINS", 3, "MODEL.ROI_CUBE_HEAD.Z_TYPE", "clusters"])
    rh = model.roi_heads
    K = rh.num_classes
    assert rh.cube_head.bbox_3D_center_depth.out_features == 3 * K
    assert tuple(rh.priors_z_scales.shape) == (K, 3) and tuple(rh.priors_z_stats.shape) == (K, 3, 2)
    with torch.no_grad():
        rh.priors_z_scales.copy_(torch.tensor([40.0, 120.0, 300.0]).expand(K, 3))
        rh.priors_z_stats.copy_(torch.tensor([[3.0, 0.8], [6.0, 1.5], [12.0, 3.0]]).expand(K, 3, 2))
    step = solver.TrainStep(cfg, model, opt, world_size=1)
    with d2.EventStorage(0):
        step(syn2.make_batch(2, 3))
        step(syn2.make_batch(2, 4))
        rep = step.report()
    assert rep["iterations_explode"] == 0 and rep["total_loss"] == rep["total_loss"]
    model.eval()
    model.roi_heads.box_predictor.test_score_thresh = 0.0
    with torch.no_grad():
        out = model(syn2.make_batch(2, 5, with_gt=False))
    z = torch.cat([o["instances"].pred_center_cam[:, 2] for o in out])
    assert len(z) > 10 and bool(torch.isfinite(z).all()) and float(z.min()) >= 0.0        # scaled sigmoid between (mu - 3 sd).clip(0) and mu + 3 sd
    with pytest.raises(ValueError, match="more than 1 cluster bin"):
        bt.build(DEV, extra=["MODEL.ROI_CUBE_HEAD.Z_TYPE", "clusters"])
