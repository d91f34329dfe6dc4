"""Winograd F(2x2, 3x3) route of the grouped 3x3 convolutions (csrc/winograd.hip + cr_gemm_batched_f32) against a float64
torch convolution: forward (bias, ReLU) and backward-data (accumulate), one shared weight over several maps (the RPN head)
and a single map; the autograd op takes it by itself in float32 and gives the gradients of the direct route."""
import importlib
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ops = importlib.import_module("3dod_amd.hipops")
DEV = "cuda:0"
f64 = torch.float64


def make(sizes, C, O, seed):
    g = torch.Generator().manual_seed(seed)
    xs = [(torch.randn(2, h, w, C, generator=g) * 0.7).to(DEV) for h, w in sizes]
    w = (torch.randn(O, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
    b = (torch.randn(O, generator=g) * 0.1).to(DEV)
    return xs, w, b


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("sizes,C,O", [([(16, 24), (8, 8), (2, 6)], 128, 128), ([(32, 32)], 128, 256), ([(12, 20), (6, 10)], 256, 128)])
def test_forward_and_backward_data_against_float64(sizes, C, O):
    xs, w, b = make(sizes, C, O, seed=len(sizes) * 100 + C)
    for relu in (False, True):
        ys = [torch.empty(x.shape[0], x.shape[1], x.shape[2], O, device=DEV) for x in xs]
        ops.wino_conv3x3_group(xs, w, ys, b, relu, None, False)
        for x, y in zip(xs, ys):
            r = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).to(f64), w.to(f64), b.to(f64), padding=1)
            r = (r.relu() if relu else r).permute(0, 2, 3, 1)
            assert rel(y, r) < 3e-6
    gs = [torch.randn(x.shape[0], x.shape[1], x.shape[2], O, device=DEV) for x in xs]
    accs = [torch.randn_like(x) for x in xs]
    accs[-1] = None
    dxs = [torch.empty_like(x) for x in xs]
    ops.wino_conv3x3_group(gs, w, dxs, None, False, accs, True)
    for g, a, dx in zip(gs, accs, dxs):
        r = torch.nn.functional.conv_transpose2d(g.permute(0, 3, 1, 2).to(f64), w.to(f64), padding=1).permute(0, 2, 3, 1)
        if a is not None:
            r = r + a.to(f64)
        assert rel(dx, r) < 3e-6


def test_autograd_op_takes_the_route_and_matches_the_direct_one(monkeypatch):
    """conv_bias_act_group in float32: same outputs and gradients with CR_WINOGRAD on (shared weight: one pipeline; separate
    weights: the big maps) and off, to the float32 rounding of two different summation orders"""
    if ops.precision() != "fp32":
        pytest.skip("float32 route")
    sizes = [(64, 64), (32, 32), (16, 16)]
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CR_WINOGRAD", mode)
        monkeypatch.setattr(ops, "WINO_MIN_TILES", 1024)
        for shared in (True, False):
            xs, w, b = make(sizes, 128, 128, seed=7)
            xs = [x.requires_grad_() for x in xs]
            ws = [w.clone().requires_grad_() for _ in sizes]
            bs = [b.clone().requires_grad_() for _ in sizes]
            if shared:
                ws, bs = [ws[0]] * 3, [bs[0]] * 3
            ys = ops.conv_bias_act_group(xs, ws, bs, pad=1, relu=True)
            sum((y * torch.linspace(-1, 1, y.numel(), device=DEV).view_as(y)).sum() for y in ys).backward()
            res[(mode, shared)] = [y.detach() for y in ys] + [x.grad for x in xs] + [ws[0].grad, bs[0].grad]
    for shared in (True, False):
        for a, b_ in zip(res[("0", shared)], res[("1", shared)]):
            assert float((a - b_).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-6


def test_odd_maps_stay_on_the_direct_route():
    xs, w, b = make([(15, 16), (8, 8)], 128, 128, seed=3)
    assert not ops.wino_supported(xs, w, 3, 1)
    assert ops.wino_supported(xs[1:], w, 3, 1) == (os.environ.get("CR_WINOGRAD", "1") == "1" and ops.precision() == "fp32")
