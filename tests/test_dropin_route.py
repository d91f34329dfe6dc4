"""The documented zero-edit route: with `<repo>/3dod_amd` on PYTHONPATH the reference's own
import statements (tools/train_net.py:39-59, demo/demo.py:22-27 of the reference) resolve to this build, every module
exists once (the top-level names are aliases of `3dod_amd.<...>`), the model builds from configs/cubercnn_DLA34_FPN.yaml
through the registries and its state dict carries the reference's key names (cubercnn/modeling/meta_arch/rcnn3d.py:894-903,
dla.py:452-458, cube_head.py:113-149, roi_heads.py:2027-2052).  Runs in a fresh interpreter, on the CPU."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import sys
    assert not any(p.rstrip("/").endswith("repo") and p != PKG for p in sys.path[:1])
    # ---- the reference's import block, verbatim names (tools/train_net.py:39-59)
    from cubercnn.solver import build_optimizer, freeze_bn, PeriodicCheckpointerOnlyOne
    from cubercnn.config import get_cfg_defaults
    from cubercnn.data import (load_omni3d_json, DatasetMapper3D, build_detection_train_loader, build_detection_test_loader,
                               get_omni3d_categories, simple_register)
    from cubercnn.evaluation import Omni3DEvaluator, Omni3Deval, Omni3DEvaluationHelper, inference_on_dataset
    from cubercnn.modeling.proposal_generator import RPNWithIgnore
    from cubercnn.modeling.roi_heads import ROIHeads3D
    from cubercnn.modeling.meta_arch import RCNN3D, build_model
    from cubercnn.modeling.backbone import build_dla_from_vision_fpn_backbone
    from cubercnn import util, vis, data
    import cubercnn.vis.logperf as utils_logperf
    import cubercnn.modeling.meta_arch
    # ---- demo/demo.py:22-27 and the proposal method's modules
    from cubercnn.modeling.meta_arch import build_model as bm2
    from ProposalNetwork.utils.spaces import Cubes
    from ProposalNetwork.proposals.proposals import propose
    from ProposalNetwork.scoring.scorefunction import score_iou, score_dimensions
    # ---- one module object per module: aliases, not copies
    import importlib
    assert sys.modules["cubercnn"] is sys.modules["3dod_amd.cubercnn"]
    assert sys.modules["cubercnn.modeling.meta_arch"] is importlib.import_module("3dod_amd.cubercnn.modeling.meta_arch")
    assert RCNN3D is importlib.import_module("3dod_amd.cubercnn.modeling").RCNN3D
    import d2lite
    assert d2lite is importlib.import_module("3dod_amd.d2lite")
    assert d2lite.META_ARCH_REGISTRY.get("RCNN3D") is RCNN3D and d2lite.ROI_HEADS_REGISTRY.get("ROIHeads3D") is ROIHeads3D
    assert d2lite.PROPOSAL_GENERATOR_REGISTRY.get("RPNWithIgnore") is RPNWithIgnore
    # ---- config + model through the registries, reference key names
    import os, torch
    cfg = d2lite.get_cfg()
    get_cfg_defaults(cfg)
    cfg.merge_from_file(os.path.join(REPO, "configs", "cubercnn_DLA34_FPN.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu"])
    torch.manual_seed(0)
    model = build_model(cfg)
    assert type(model).__name__ == cfg.MODEL.META_ARCHITECTURE == "RCNN3D"
    sd = model.state_dict()
    for k in ("backbone.bottom_up.base_layer.0.weight", "backbone.bottom_up.level2.tree1.conv1.weight",
              "backbone.bottom_up.level5.root.conv.weight", "backbone.fpn_lateral2.weight", "backbone.fpn_output5.bias",
              "proposal_generator.rpn_head.conv.weight", "proposal_generator.rpn_head.objectness_logits.weight",
              "proposal_generator.rpn_head.anchor_deltas.bias", "roi_heads.box_head.fc1.weight",
              "roi_heads.box_predictor.cls_score.weight", "roi_heads.box_predictor.bbox_pred.bias",
              "roi_heads.cube_head.feature_generator.fc1.weight", "roi_heads.cube_head.bbox_3D_dims.weight",
              "roi_heads.cube_head.bbox_3D_center_deltas.bias", "roi_heads.cube_head.bbox_3D_pose.weight",
              "roi_heads.cube_head.bbox_3D_center_depth.bias", "roi_heads.cube_head.bbox_3D_uncertainty.weight",
              "roi_heads.priors_dims_per_cat"):
        assert k in sd, k
    assert tuple(sd["backbone.bottom_up.level2.tree1.conv1.weight"].shape) == (64, 32, 3, 3)      # logical KCRS shape
    assert tuple(sd["roi_heads.cube_head.feature_generator.fc1.weight"].shape) == (1024, 256 * 7 * 7)
    n = sum(p.numel() for p in model.parameters())
    assert n == 47908514, n                                        # SURVEY 8(e): 47.9 M parameters
    # a checkpoint with the reference's names loads (strict) and changes the weights
    other = {k: torch.randn_like(v) if v.dtype.is_floating_point else v for k, v in sd.items()}
    missing = model.load_state_dict(other, strict=True)
    assert torch.equal(model.state_dict()["roi_heads.cube_head.bbox_3D_dims.weight"], other["roi_heads.cube_head.bbox_3D_dims.weight"])
    opt = build_optimizer(cfg, model)
    assert opt.flat_p.numel() >= n
    t = utils_logperf.print_ap_dataset_histogram({"Synth": {"iters": 10, "AP2D": 1.5, "AP3D": 0.5}})
    assert "Synth" in t and "AP3D" in t
    print("DROPIN-OK")
''')


def test_reference_import_names_resolve_with_one_pythonpath_entry():
    pkg = os.path.join(ROOT, "3dod_amd")
    env = dict(os.environ, PYTHONPATH=pkg, PYTHONDONTWRITEBYTECODE="1")
    code = f"PKG = {pkg!r}\nREPO = {ROOT!r}\n" + SCRIPT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd="/tmp", timeout=600)
    assert out.returncode == 0 and "DROPIN-OK" in out.stdout, out.stderr[-3000:] + out.stdout[-500:]
