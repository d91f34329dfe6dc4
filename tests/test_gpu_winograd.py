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


@pytest.mark.parametrize("sizes,C,O", [([(16, 24), (8, 8), (2, 6)], 128, 128), ([(64, 64)], 128, 256), ([(12, 20), (6, 10)], 256, 128)])
def test_weight_gradient_against_float64(sizes, C, O):
    """dW += G^T (dM^T V) G and db += sum dY (cr_wino_dy, cr_wgrad_batched_f32, cr_wino_filter_grad), accumulated INTO the sinks"""
    xs, w, b = make(sizes, C, O, seed=len(sizes) * 10 + O)
    gs = [torch.randn(x.shape[0], x.shape[1], x.shape[2], O, device=DEV) for x in xs]
    w0 = (torch.randn(O, C, 3, 3, device=DEV) * 0.5).contiguous(memory_format=torch.channels_last)
    b0 = torch.randn(O, device=DEV)
    sink, bsink = w0.clone(memory_format=torch.preserve_format), b0.clone()
    ops.wino_wgrad_group(gs, xs, sink, bsink)
    refw = sum(torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).to(f64), w.shape, g.permute(0, 3, 1, 2).to(f64), padding=1)
               for x, g in zip(xs, gs))
    refb = sum(g.to(f64).sum((0, 1, 2)) for g in gs)
    assert float(((sink - w0).double() - refw).abs().max() / refw.abs().max()) < 5e-6
    assert float(((bsink - b0).double() - refb).abs().max() / refb.abs().max()) < 5e-6
    sink2 = torch.zeros_like(sink)
    ops.wino_wgrad_group(gs, xs, sink2, None)                       # no bias sink: weights only
    assert rel(sink2, refw) < 5e-6


def test_autograd_weight_gradient_into_sinks(monkeypatch):
    """parameters with gradient sinks (what FlatSGD attaches): the op's weight / bias gradients land in the sinks, Winograd route
    and direct route (CR_WINO_WGRAD=0) agreeing to float32 rounding; CR_DETERMINISTIC=1 keeps the direct kernels"""
    if ops.precision() != "fp32":
        pytest.skip("float32 route")
    monkeypatch.setattr(ops, "WINO_MIN_TILES", 1024)
    sizes = [(64, 64), (32, 32), (16, 16)]
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CR_WINO_WGRAD", mode)
        assert ops.wino_wgrad_on() == (mode == "1")
        for shared in (True, False):
            xs, w, b = make(sizes, 128, 128, seed=11)
            ws = [w.clone().requires_grad_() for _ in sizes]
            bs = [b.clone().requires_grad_() for _ in sizes]
            if shared:
                ws, bs = [ws[0]] * 3, [bs[0]] * 3
            for p in set(ws) | set(bs):
                p._cr_grad = torch.zeros_like(p, memory_format=torch.preserve_format)
            ys = ops.conv_bias_act_group(xs, ws, bs, pad=1, relu=True)
            sum((y * torch.linspace(-1, 1, y.numel(), device=DEV).view_as(y)).sum() for y in ys).backward()
            assert all(p.grad is None for p in ws + bs)
            res[(mode, shared)] = [p._cr_grad for p in ws + bs]
    for shared in (True, False):
        for a, b_ in zip(res[("0", shared)], res[("1", shared)]):
            assert float(a.abs().max()) > 0
            assert float((a - b_).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-6
    # the forward pass kept B^T x B for the weight gradient (CR_WINO_KEEP_V, default on): bit-equal to transforming again
    monkeypatch.setenv("CR_WINO_WGRAD", "1")
    monkeypatch.setenv("CR_WINO_KEEP_V", "0")
    xs, w, b = make(sizes, 128, 128, seed=11)
    w.requires_grad_()
    w._cr_grad = torch.zeros_like(w, memory_format=torch.preserve_format)
    ys = ops.conv_bias_act_group(xs, [w] * 3, [b] * 3, pad=1, relu=True)
    sum((y * torch.linspace(-1, 1, y.numel(), device=DEV).view_as(y)).sum() for y in ys).backward()
    # atomics order differs run to run: float32 rounding only
    assert float((w._cr_grad - res[("1", True)][0]).abs().max()) <= 2e-5 * float(w._cr_grad.abs().max())
    monkeypatch.setenv("CR_DETERMINISTIC", "1")
    assert not ops.wino_wgrad_on()
