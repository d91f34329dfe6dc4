"""GPU: the ResNet34-FPN variant of Cube R-CNN (configs/cubercnn_ResNet34_FPN.yaml) on the HIP kernels.
torchvision is absent, so the trunk's structure is restated from its public definition ("parity unpinned" w.r.t.
torchvision); what IS checked: state-dict keys of torchvision's resnet34, every residual block in isolation against the
bf16-emulating float32 oracle on the block's actual GPU inputs, pyramid shapes, and that the full train step learns."""
import importlib
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    syn = importlib.import_module("3dod_amd.synthetic")
    modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
    solver = importlib.import_module("3dod_amd.cubercnn.solver")
    cfg = syn.make_cfg(os.path.join(ROOT, "configs", "cubercnn_ResNet34_FPN.yaml"),
                       overrides=["MODEL.DEVICE", "cuda:0", "VIS_PERIOD", 0, "log", False, "SOLVER.BASE_LR", 0.0025])
    torch.manual_seed(0)
    model = modeling.build_model(cfg).to(DEV).train()
    opt = solver.build_optimizer(cfg, model)
    return cfg, model, opt, syn, solver


def _rel(a, b):
    a = a.float().cpu(); b = b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12)), float((a - b).abs().max() / (b.abs().max() + 1e-12))


def test_state_dict_keys_and_pyramid(built):
    cfg, model, opt, syn, solver = built
    keys = set(model.state_dict().keys())
    for k in ("backbone.bottom_up.conv1.weight", "backbone.bottom_up.bn1.running_mean",
              "backbone.bottom_up.layer1.2.conv2.weight", "backbone.bottom_up.layer2.0.downsample.0.weight",
              "backbone.bottom_up.layer2.0.downsample.1.weight", "backbone.bottom_up.layer3.5.bn2.bias",
              "backbone.bottom_up.layer4.2.conv1.weight", "backbone.fpn_lateral2.weight", "backbone.fpn_output6.bias"):
        assert k in keys, k
    bu = [k for k in keys if k.startswith("backbone.bottom_up.") and k.endswith("conv1.weight") or ".conv2.weight" in k]
    assert sum(1 for k in keys if k.startswith("backbone.bottom_up.layer") and k.endswith(".conv1.weight")) == 16   # 3+4+6+3
    n_params = sum(p.numel() for n, p in model.backbone.bottom_up.named_parameters())
    assert n_params == 21284672                                                  # torchvision resnet34 without fc
    batch = syn.make_batch(2, 3, with_gt=False)
    with torch.no_grad():
        images, x = model.preprocess_image(batch)
        feats = model.backbone(x)
    # detectron2 FPN + LastLevelMaxPool: "p7" = max_pool2d(k=1, s=2) of the BOTTOM-UP p5 (its in_feature exists there)
    assert list(feats.keys()) == ["p2", "p3", "p4", "p5", "p6", "p7"]
    for name, s, c in (("p2", 128, 256), ("p3", 64, 256), ("p4", 32, 256), ("p5", 16, 256), ("p6", 8, 256), ("p7", 8, 512)):
        assert tuple(feats[name].shape) == (2, s, s, c), (name, feats[name].shape)


def test_blocks_in_isolation(built, precision):
    """fp32 (reference precision): relative L2 <= 2e-5 / max-norm <= 2e-4 against the float32 oracle; bf16 fast mode:
    2e-3 / 2e-2 against the bf16-emulating oracle"""
    from oracle import cpu_backend
    emu = precision == "bf16"
    t_l2, t_mx = (2e-3, 2e-2) if emu else (2e-5, 2e-4)
    cfg, model, opt, syn, solver = built
    resnet = importlib.import_module("3dod_amd.cubercnn.modeling.backbone.resnet")
    modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
    batch = syn.make_batch(2, 21, with_gt=False)
    rec = {}

    def mk(name):
        def hook(m, inp, out):
            rec[name] = ([i.detach().float().cpu() for i in inp if torch.is_tensor(i)], out.detach().float().cpu())
        return hook
    bu = model.backbone.bottom_up
    hooks = [m.register_forward_hook(mk(n)) for n, m in bu.named_modules() if isinstance(m, resnet.BasicBlock)]
    stem = {}
    with torch.no_grad():
        images, x = model.preprocess_image(batch)
        stem["x"] = x.float().cpu()
        s1 = resnet._conv_bn(x, bu.conv1, bu.bn1, relu=True)
        stem["conv"] = s1.float().cpu()
        stem["pool"] = importlib.import_module("3dod_amd.hipops").maxpool3x3s2(s1).float().cpu()
        model.backbone(x)
    for h in hooks:
        h.remove()
    assert len(rec) == 16
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    saved = {n: importlib.import_module(n).ops for n in cpu_backend.PATCHED}
    try:
        cpu_backend.install()
        cpu_backend.EMULATE_BF16 = emu
        ref = modeling.build_model(syn.make_cfg(os.path.join(ROOT, "configs", "cubercnn_ResNet34_FPN.yaml"),
                                                overrides=["MODEL.DEVICE", "cpu", "VIS_PERIOD", 0, "log", False]))
        ref.load_state_dict(sd)
        ref.train()
        rbu = ref.backbone.bottom_up
        mods = dict(rbu.named_modules())
        with torch.no_grad():
            c = resnet._conv_bn(stem["x"], rbu.conv1, rbu.bn1, relu=True)
            l2e, mx = _rel(stem["conv"], c)
            assert l2e < t_l2 and mx < t_mx, ("stem conv", l2e, mx)
            l2e, mx = _rel(stem["pool"], cpu_backend.maxpool3x3s2(stem["conv"]))
            assert l2e == 0.0, ("stem pool", l2e, mx)
            for name, (inp, out) in rec.items():
                l2e, mx = _rel(out, mods[name](*inp))
                assert l2e < t_l2 and mx < t_mx, (name, l2e, mx)
    finally:
        cpu_backend.EMULATE_BF16 = False
        for n, o in saved.items():
            importlib.import_module(n).ops = o


def test_resnet_train_steps_learn(built):
    cfg, model, opt, syn, solver = built
    d2 = importlib.import_module("3dod_amd.d2lite")
    step = solver.TrainStep(cfg, model, opt, world_size=1)
    batch = syn.make_batch(2, 7)
    totals = []
    with d2.EventStorage(0):
        for _ in range(8):
            step(batch)
            rep = step.report()
            totals.append(rep["total_loss"])
    assert all(t == t and abs(t) < 1e4 for t in totals), totals
    assert rep["iterations_explode"] == 0, rep
    assert min(totals[-3:]) < totals[0], totals
