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


def test_cube_decode_infer_kernel_matches_reference_eval_golden(golden_dir):
    """cr_cube_decode_infer (the fused inference decode of the 3D head, roi_heads.py:2353-2436,2682-2735) against
    tests/golden/cubehead_eval.npz = the reference's own ROIHeads3D._forward_cube in eval mode: corners, centres,
    dimensions, pose and merged scores within 1e-4 relative (north_star's corner tolerance)."""
    import os
    import numpy as np
    util = importlib.import_module("3dod_amd.cubercnn.util.math_util")
    g = np.load(os.path.join(golden_dir, "cubehead_eval.npz"), allow_pickle=False)
    T = lambda k: torch.tensor(g[k]).to(DEV)
    n, K = g["in_z"].shape[0], g["in_z"].shape[1]
    ld = (13 * K + 15) // 16 * 16
    raw = torch.zeros((n, ld), device=DEV)
    raw[:, 0:2 * K] = T("in_deltas").reshape(n, -1)
    raw[:, 2 * K:5 * K] = T("in_dims").reshape(n, -1)
    raw[:, 5 * K:11 * K] = T("in_pose6").reshape(n, -1)
    raw[:, 11 * K:12 * K] = T("in_z").reshape(n, -1)
    raw[:, 12 * K:13 * K] = T("in_uncert")
    layout = (0, 2 * K, 5 * K, 11 * K, 12 * K)
    n_per = g["n_per"].tolist()
    img = torch.repeat_interleave(torch.arange(len(n_per)), torch.tensor(n_per)).to(DEV)
    rows = []
    for k, r in zip(g["Ks"], g["ratios"]):
        r = float(r)
        v2r = util.compute_virtual_scale_from_focal_spaces(float(k[1, 1]), 512.0 * r, 512.0, 512.0)
        rows.append([float(k[0, 0]) / r, float(k[1, 1]) / r, float(k[0, 2]) / r, float(k[1, 2]) / r, float(v2r), r])
    meta6 = torch.tensor(rows, dtype=torch.float32, device=DEV)
    priors = T("priors")[0, :, 0, :].contiguous()
    o = ops.cube_decode_infer(raw, layout, K, T("classes"), img, T("pred_boxes"), meta6, priors, allocentric=True).cpu().numpy()
    chk = lambda got, key: np.testing.assert_allclose(got, g[key], rtol=1e-4, atol=1e-5, err_msg=key)
    chk(o[:, 18:42].reshape(n, 8, 3), "out_pred_bbox3D")
    chk(o[:, 0:3], "out_pred_center_cam")
    chk(o[:, 6:8], "out_pred_center_2D")
    chk(o[:, 3:6], "out_pred_dimensions")
    chk(o[:, 9:18].reshape(n, 3, 3), "out_pred_pose")
    chk(np.sqrt(g["scores_2d"] * o[:, 8]), "out_scores")
