"""Golden vectors for the Cube R-CNN 3D head: runs the REFERENCE's own
ROIHeads3D._forward_cube (cubercnn/modeling/roi_heads/roi_heads.py:2237-2735) on CPU, in training and
in eval mode, with the pooler / cube head replaced by given tensors, and records losses, their gradients
w.r.t. the head outputs, and the decoded instances.

Third-party symbols the method touches are stood in by this repo's d2lite structures and math_util
transforms (detectron2 / pytorch3d are not installed): Instances, Boxes, select_foreground_proposals,
get_event_storage, axis_angle_to_matrix.  Everything else (decode, virtual depth, allocentric pose,
disentangled corner sets, chamfer, uncertainty weighting, safely_reduce_losses) is the reference's code.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_cubehead.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _refimport  # noqa: E402

_refimport.install()
d2 = importlib.import_module("3dod_amd.d2lite")
my_util = importlib.import_module("3dod_amd.cubercnn.util.math_util")
my_rh = importlib.import_module("3dod_amd.cubercnn.modeling.roi_heads.roi_heads")
syn = importlib.import_module("3dod_amd.synthetic")

from cubercnn.util import math_util as ref_math          # reference
ref_math.axis_angle_to_matrix = my_util.axis_angle_to_matrix
import cubercnn.modeling.roi_heads.roi_heads as ref_rh   # reference (detectron2 names are stubs here)

ref_rh.Instances = d2.Instances
ref_rh.Boxes = d2.Boxes
ref_rh.select_foreground_proposals = my_rh.select_foreground_proposals
storage = d2.EventStorage(0)
ref_rh.get_event_storage = lambda: storage
torch.set_num_threads(1)


def make_case(seed, training, K_classes=50):
    g = torch.Generator().manual_seed(seed)
    batch = syn.make_batch(3, seed)
    # make the three images differ in scale ratio and K like real resized inputs
    ratios = [1.0, 1.25, 0.8]
    instances, Ks, head = [], [], {}
    n_per = [7, 5, 9]
    for i, (b, n) in enumerate(zip(batch, n_per)):
        gt = b["instances"]
        idx = torch.randint(0, len(gt), (n,), generator=g)
        inst = d2.Instances((512, 512))
        jit = torch.randn(n, 4, generator=g) * 6
        inst.proposal_boxes = d2.Boxes(gt.gt_boxes.tensor[idx] + jit)
        inst.pred_boxes = d2.Boxes(gt.gt_boxes.tensor[idx] + torch.randn(n, 4, generator=g) * 3)
        if training:
            inst.gt_classes = gt.gt_classes[idx]
            inst.gt_boxes3D = gt.gt_boxes3D[idx]
            inst.gt_poses = gt.gt_poses[idx]
        else:
            inst.pred_classes = gt.gt_classes[idx]
            inst.scores = torch.rand(n, generator=g)
        instances.append(inst)
        Ks.append(torch.tensor(b["K"]))
    n = sum(n_per)
    head["deltas"] = torch.randn(n, K_classes, 2, generator=g) * 0.1
    head["z"] = torch.randn(n, K_classes, 1, generator=g) * 0.5 + 3.0
    head["dims"] = torch.randn(n, K_classes, 3, generator=g) * 0.3
    head["pose6"] = torch.randn(n, K_classes, 6, generator=g)
    head["uncert"] = (torch.randn(n, K_classes, generator=g) * 0.5 + 1.0).clip(0.01)
    priors = torch.rand(1, K_classes, 2, 3, generator=g) * 0.8 + 0.3
    return instances, Ks, ratios, head, priors


def run(seed, training, z_type="direct", cluster_bins=1):
    instances, Ks, ratios, head, priors = make_case(seed, training)
    K_classes = head["z"].shape[1]
    z_scales = z_stats = None
    if cluster_bins > 1:
        # the depth predictor has one output per (bin, class): (n, bins, K, 1) (cube_head.py:196-197); scale centres / depth
        # statistics per (class, bin) like priors['priors_bins'] (roi_heads.py:2032-2051)
        g2 = torch.Generator().manual_seed(seed + 100)
        n = head["z"].shape[0]
        head["z"] = torch.randn(n, cluster_bins, K_classes, 1, generator=g2) * 0.5 + (0.0 if z_type == "clusters" else 3.0)
        z_scales = torch.sort(torch.rand(K_classes, cluster_bins, generator=g2) * 250 + 20, dim=1).values
        z_stats = torch.stack((torch.rand(K_classes, cluster_bins, generator=g2) * 8 + 2,
                               torch.rand(K_classes, cluster_bins, generator=g2) * 1.5 + 0.3), dim=-1)
    if z_type == "log":
        head["z"] = head["z"] - 1.5                      # exp(1.5 +- 0.5): a few metres
    elif z_type == "sigmoid":
        head["z"] = head["z"] - 6.0                      # 100 sigmoid(-3 +- 0.5): a few metres
    leaves = {k: v.clone().requires_grad_(training) for k, v in head.items()}
    pose = my_util.rotation_6d_to_matrix(leaves["pose6"].view(-1, 6)).view(leaves["pose6"].shape[0], -1, 3, 3)
    n = leaves["z"].shape[0]
    self = types.SimpleNamespace()
    cfgv = dict(in_features=["p2"], training=training, num_classes=50, scale_roi_boxes=0.0, virtual_depth=True,
                virtual_focal=512.0, cluster_bins=cluster_bins, use_confidence=1.0, dims_priors_enabled=True,
                dims_priors_func="exp", allocentric_pose=True, z_type=z_type, disentangled_loss=True,
                chamfer_pose=True, loss_w_3d=1.0, loss_w_xy=1.0, loss_w_z=1.0, loss_w_dims=20.0, loss_w_pose=7.0,
                loss_w_joint=1.0, inverse_z_weight=False)
    for k, v in cfgv.items():
        setattr(self, k, v)
    self.priors_dims_per_cat = priors
    if cluster_bins > 1:
        self.priors_z_scales = z_scales
        self.priors_z_stats = z_stats
    self.cube_pooler = lambda feats, boxes: torch.zeros(n, 4)
    self.cube_head = lambda x: (leaves["deltas"], leaves["z"], leaves["dims"], pose, leaves["uncert"])
    C = ref_rh.ROIHeads3D
    for name in ("l1_loss", "chamfer_loss", "scale_proposals", "safely_reduce_losses"):
        setattr(self, name, types.MethodType(getattr(C, name), self))
    im_dims = [(512, 512)] * 3
    out = C._forward_cube(self, {"p2": None}, instances, Ks, im_dims, ratios)
    rec = {}
    for k, v in head.items():
        rec["in_" + k] = v.numpy()
    rec["priors"] = priors.numpy()
    if cluster_bins > 1:
        rec["priors_z_scales"], rec["priors_z_stats"] = z_scales.numpy(), z_stats.numpy()
    rec["ratios"] = np.array(ratios, np.float32)
    rec["Ks"] = torch.stack(Ks).numpy()
    rec["n_per"] = np.array([len(i) for i in instances])
    rec["proposal_boxes"] = torch.cat([i.proposal_boxes.tensor for i in instances]).numpy()
    rec["pred_boxes"] = torch.cat([i.pred_boxes.tensor for i in instances]).numpy()
    if training:
        pred_instances, losses = out
        rec["gt_classes"] = torch.cat([i.gt_classes for i in instances]).numpy()
        rec["gt_boxes3D"] = torch.cat([i.gt_boxes3D for i in instances]).numpy()
        rec["gt_poses"] = torch.cat([i.gt_poses for i in instances]).numpy()
        total = sum(losses.values())
        total.backward()
        for k, v in losses.items():
            rec["loss_" + k.replace("/", "_")] = v.detach().numpy()
        for k, v in leaves.items():
            rec["grad_" + k] = v.grad.numpy()
    else:
        pred_instances = out
        rec["classes"] = torch.cat([i.pred_classes for i in instances]).numpy()
        rec["scores_in"] = None
    for f in ("pred_bbox3D", "pred_center_cam", "pred_center_2D", "pred_dimensions", "pred_pose", "scores"):
        rec["out_" + f] = torch.cat([i.get(f) for i in pred_instances]).detach().numpy()
    rec = {k: v for k, v in rec.items() if v is not None}
    rec["notes"] = ("reference ROIHeads3D._forward_cube (roi_heads.py:2237-2735), training=%s; third-party symbols stood "
                    "in (Instances/Boxes/select_foreground_proposals/event storage/axis_angle_to_matrix/"
                    "rotation_6d_to_matrix): parity unpinned for those, pinned for the reference's own arithmetic" % training)
    return rec


if __name__ == "__main__":
    tr = run(11, True)
    # eval: the incoming 2D scores are part of the input
    instances, *_ = make_case(12, False)
    ev = run(12, False)
    ev["scores_2d"] = torch.cat([i.scores for i in make_case(12, False)[0]]).numpy()
    np.savez_compressed(os.path.join(HERE, "cubehead_train.npz"), **tr)
    np.savez_compressed(os.path.join(HERE, "cubehead_eval.npz"), **ev)
    for zt in ("sigmoid", "log"):                        # MODEL.ROI_CUBE_HEAD.Z_TYPE variants (roi_heads.py:2404-2410)
        t2 = run(13, True, zt)
        e2 = run(14, False, zt)
        e2["scores_2d"] = torch.cat([i.scores for i in make_case(14, False)[0]]).numpy()
        np.savez_compressed(os.path.join(HERE, "cubehead_train_z%s.npz" % zt), **t2)
        np.savez_compressed(os.path.join(HERE, "cubehead_eval_z%s.npz" % zt), **e2)
        print(zt, {k: float(v) for k, v in t2.items() if k.startswith("loss_")})
    for zt in ("direct", "clusters"):                    # CLUSTER_BINS = 3 (roi_heads.py:2343-2356), with and without cluster depth priors
        t3 = run(15, True, zt, cluster_bins=3)
        e3 = run(16, False, zt, cluster_bins=3)
        e3["scores_2d"] = torch.cat([i.scores for i in make_case(16, False)[0]]).numpy()
        np.savez_compressed(os.path.join(HERE, "cubehead_train_bins3_%s.npz" % zt), **t3)
        np.savez_compressed(os.path.join(HERE, "cubehead_eval_bins3_%s.npz" % zt), **e3)
        print("bins3", zt, {k: float(v) for k, v in t3.items() if k.startswith("loss_")})
    for k in sorted(tr):
        if k.startswith("loss_"):
            print(k, float(tr[k]))
    print({k: v.shape for k, v in ev.items() if hasattr(v, "shape") and k.startswith("out_")})
