"""CPU: the torch-expression statement of ROIHeads3D._forward_cube (decode + disentangled corner losses; oracle/cube_list.py,
attached to the head by oracle.list_path.install_heads) against golden vectors produced by the REFERENCE's own method
(tests/golden/make_golden_cubehead.py).  The pooler and the cube head are replaced by the fixture's tensors, so no GPU
kernel is involved -- this pins the arithmetic the fused kernels are then compared with on the GPU."""
import importlib
import os

import numpy as np
import pytest
import torch

d2 = importlib.import_module("3dod_amd.d2lite")
syn = importlib.import_module("3dod_amd.synthetic")
modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
util = importlib.import_module("3dod_amd.cubercnn.util.math_util")


@pytest.fixture(scope="module")
def heads():
    cfg = syn.make_cfg(overrides=["MODEL.DEVICE", "cpu", "VIS_PERIOD", 0, "log", False])
    shapes = {f"p{l}": d2.ShapeSpec(channels=256, stride=2 ** l) for l in range(2, 7)}
    from oracle import list_path
    return list_path.install_heads(modeling.build_roi_heads(cfg, shapes))


def _clusters(heads, g):
    """CLUSTER_BINS and the cluster tables of a fixture (priors_z_scales (K,bins), priors_z_stats (K,bins,2)), or back to one bin"""
    if "priors_z_scales" in g.files:
        heads.cluster_bins = int(g["priors_z_scales"].shape[1])
        heads.priors_z_scales = torch.nn.Parameter(torch.tensor(g["priors_z_scales"]))
        heads.priors_z_stats = torch.nn.Parameter(torch.tensor(g["priors_z_stats"]))
    else:
        heads.cluster_bins = 1


def _setup(heads, g, training):
    n_per = g["n_per"].tolist()
    T = lambda k: torch.tensor(g[k])
    split = lambda t: t.split(n_per)
    insts = []
    for i in range(len(n_per)):
        inst = d2.Instances((512, 512))
        inst.proposal_boxes = d2.Boxes(split(T("proposal_boxes"))[i])
        inst.pred_boxes = d2.Boxes(split(T("pred_boxes"))[i])
        if training:
            inst.gt_classes = split(T("gt_classes"))[i]
            inst.gt_boxes3D = split(T("gt_boxes3D"))[i]
            inst.gt_poses = split(T("gt_poses"))[i]
        else:
            inst.pred_classes = split(T("classes"))[i]
            inst.scores = split(T("scores_2d"))[i]
        insts.append(inst)
    leaves = {k: T("in_" + k).requires_grad_(training) for k in ("deltas", "z", "dims", "pose6", "uncert")}
    n = leaves["z"].shape[0]
    pose = util.rotation_6d_to_matrix(leaves["pose6"].view(-1, 6)).view(n, -1, 3, 3)
    heads.priors_dims_per_cat.data = T("priors")
    class _Fake(torch.nn.Module):
        def __init__(self, fn):
            super().__init__()
            self.fn = fn

        def forward(self, *a):
            return self.fn(*a)
    heads.cube_pooler = _Fake(lambda feats, boxes: torch.zeros(n, 4))
    heads.cube_head = _Fake(lambda x: (leaves["deltas"], leaves["z"], leaves["dims"], pose, leaves["uncert"]))
    Ks = [torch.tensor(k) for k in g["Ks"]]
    return insts, leaves, Ks, [float(r) for r in g["ratios"]]


# MODEL.ROI_CUBE_HEAD.Z_TYPE (roi_heads.py:2404-2436) and CLUSTER_BINS = 3 (:2343-2356)
ZT = [("direct", ""), ("sigmoid", "_zsigmoid"), ("log", "_zlog"), ("direct", "_bins3_direct"), ("clusters", "_bins3_clusters")]


@pytest.mark.parametrize("z_type,suffix", ZT)
def test_forward_cube_training_matches_reference(heads, golden_dir, z_type, suffix):
    g = np.load(os.path.join(golden_dir, "cubehead_train%s.npz" % suffix), allow_pickle=False)
    heads.z_type = z_type
    _clusters(heads, g)
    heads.train()
    insts, leaves, Ks, ratios = _setup(heads, g, True)
    with d2.EventStorage(0):
        pred, losses = heads._forward_cube({f: None for f in heads.in_features}, insts, Ks, [(512, 512)] * 3, ratios)
    for k, v in losses.items():
        ref = float(g["loss_" + k.replace("/", "_")])
        assert abs(float(v) - ref) <= 1e-5 * max(1.0, abs(ref)), (k, float(v), ref)
    assert set(losses) == {"Cube/" + s for s in ("uncert", "loss_dims", "loss_xy", "loss_z", "loss_pose", "loss_joint")}
    sum(losses.values()).backward()
    for k, leaf in leaves.items():
        np.testing.assert_allclose(leaf.grad.numpy(), g["grad_" + k], rtol=2e-4, atol=2e-6, err_msg=k)
    for f in ("pred_bbox3D", "pred_center_cam", "pred_center_2D", "pred_dimensions", "pred_pose", "scores"):
        got = torch.cat([i.get(f) for i in pred]).detach().numpy()
        np.testing.assert_allclose(got, g["out_" + f], rtol=1e-4, atol=1e-5, err_msg=f)
    heads.z_type, heads.cluster_bins = "direct", 1


@pytest.mark.parametrize("z_type,suffix", ZT)
def test_forward_cube_eval_matches_reference(heads, golden_dir, z_type, suffix):
    g = np.load(os.path.join(golden_dir, "cubehead_eval%s.npz" % suffix), allow_pickle=False)
    heads.z_type = z_type
    _clusters(heads, g)
    heads.eval()
    insts, leaves, Ks, ratios = _setup(heads, g, False)
    with torch.no_grad():
        pred = heads._forward_cube({f: None for f in heads.in_features}, insts, Ks, [(512, 512)] * 3, ratios)
    heads.train()
    heads.z_type, heads.cluster_bins = "direct", 1
    for f in ("pred_bbox3D", "pred_center_cam", "pred_center_2D", "pred_dimensions", "pred_pose", "scores"):
        got = torch.cat([i.get(f) for i in pred]).numpy()
        np.testing.assert_allclose(got, g["out_" + f], rtol=1e-4, atol=1e-5, err_msg=f)   # north_star: corners 1e-4 rel


def test_empty_returns_instances_like_reference(heads):
    heads.eval()
    inst = d2.Instances((512, 512))
    inst.pred_boxes = d2.Boxes(torch.zeros(0, 4)); inst.pred_classes = torch.zeros(0, dtype=torch.long)
    inst.scores = torch.zeros(0)
    out = heads._forward_cube({f: None for f in heads.in_features}, [inst], [torch.eye(3)], [(512, 512)], [1.0])
    heads.train()
    assert out[0] is inst and not inst.has("pred_bbox3D")       # roi_heads.py:2278-2279 early return
