"""SOLVER.NESTEROV and SOLVER.CLIP_GRADIENTS of the flat optimizers against torch.optim.SGD + torch.nn.utils.clip_grad_*
(detectron2's maybe_add_gradient_clipping clips every parameter on its own, solver/build.py:68)."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
solver = importlib.import_module("3dod_amd.cubercnn.solver")
build = importlib.import_module("3dod_amd.cubercnn.solver.build")


def make_params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 32, 3, 3), (64,), (1024, 12544), (7,), (16, 4, 7, 7), (1, 1)]
    return [torch.randn(s, generator=g).mul_(0.1) for s in shapes]


def run_pair(clip, nesterov, steps=3):
    ref = [p.clone().to(DEV).requires_grad_() for p in make_params(0)]
    mine = [torch.nn.Parameter(p.clone().to(DEV)) for p in make_params(0)]
    if mine[0].dim() == 4:
        mine[0].data = mine[0].data.contiguous(memory_format=torch.channels_last)
    topt = torch.optim.SGD(ref, lr=0.05, momentum=0.9, weight_decay=1e-3, nesterov=nesterov)
    opt = build.FlatSGD([(p, 0.05, 1e-3) for p in mine], 0.9, nesterov)
    if clip is not None:
        opt.set_gradient_clipping(*clip)
    for it in range(steps):
        gs = [g.to(DEV) * (3.0 if it == 1 else 0.3) for g in make_params(100 + it)]
        for p, g in zip(ref, gs):
            p.grad = g.clone()
        if clip is not None:
            for p in ref:                         # per parameter, like detectron2's per-param clipper
                if clip[0] == "value":
                    torch.nn.utils.clip_grad_value_(p, clip[1])
                else:
                    torch.nn.utils.clip_grad_norm_(p, clip[1], clip[2])
        topt.step()
        opt.zero_grad()
        for p, g in zip(mine, gs):
            p._cr_grad.copy_(g)
        opt.step()
    torch.cuda.synchronize()
    for a, b in zip(ref, mine):
        torch.testing.assert_close(b.detach().contiguous(), a.detach(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("nesterov", [False, True])
def test_sgd_nesterov(nesterov):
    run_pair(None, nesterov)


@pytest.mark.parametrize("clip", [("value", 0.05, 2.0), ("norm", 1.0, 2.0), ("norm", 0.5, 1.0), ("norm", 0.02, float("inf")),
                                  ("norm", 2.0, 3.0)])
def test_gradient_clipping(clip):
    run_pair(clip, False)


def test_clipping_folds_the_gradient_scale():
    """after an all-reduce the gradient is averaged by grad_scale = 1 / world: clipping sees the averaged gradient"""
    ps = [torch.nn.Parameter(p.clone().to(DEV)) for p in make_params(1)]
    opt = build.FlatSGD([(p, 0.1, 0.0) for p in ps], 0.0)
    opt.set_gradient_clipping("norm", 1.0, 2.0)
    gs = [g.to(DEV) for g in make_params(7)]
    before = [p.detach().clone() for p in ps]
    for p, g in zip(ps, gs):
        p._cr_grad.copy_(g * 4.0)
    opt.step(grad_scale=0.25)
    for p, b, g in zip(ps, before, gs):
        coef = min(1.0, 1.0 / (float(g.norm()) + 1e-6))
        torch.testing.assert_close(p.detach().contiguous(), (b - 0.1 * coef * g).contiguous(), rtol=2e-5, atol=2e-6)


def test_build_optimizer_reads_the_config():
    cfgm = importlib.import_module("3dod_amd.cubercnn.config")
    d2 = importlib.import_module("3dod_amd.d2lite")
    cfg = d2.get_cfg()
    cfgm.get_cfg_defaults(cfg)
    cfg.SOLVER.NESTEROV = True
    cfg.SOLVER.CLIP_GRADIENTS.ENABLED = True
    cfg.SOLVER.CLIP_GRADIENTS.CLIP_TYPE = "norm"
    cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE = 0.7
    m = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 4)).to(DEV)
    opt = solver.build_optimizer(cfg, m)
    assert opt.nesterov and opt.clip == ("norm", 0.7, 2.0)
