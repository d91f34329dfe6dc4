"""Grouped convolution launches (cr_conv2d_fwd_group / _bwd_data_group / _bwd_weight_group: the five pyramid levels of the FPN
output convolutions and of the RPN head's shared convolution in one grid per direction) against the per-convolution kernels
and the float32 CPU oracle: outputs, input gradients (with a gradient-slot contribution folded into the epilogue), weight and
bias gradients accumulated into parameter sinks, shared weights, an output nobody differentiates."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu
ops = importlib.import_module("3dod_amd.hipops")
DEV = torch.device("cuda:0")
f32 = torch.float32


def _rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-20))


def _levels(g, N=2, C=128, sizes=((40, 48), (20, 24), (10, 12), (5, 6), (3, 3)), dt=f32):
    return [torch.randn(N, h, w, C, generator=g).to(DEV).to(dt) for h, w in sizes]


def _weights(g, n, C, k, shared):
    def one():
        w = (torch.randn(C, C, k, k, generator=g) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last)
        b = (torch.randn(C, generator=g) * 0.1).to(DEV)
        return w, b
    if shared:
        w, b = one()
        return [w] * n, [b] * n
    ws, bs = zip(*[one() for _ in range(n)])
    return list(ws), list(bs)


def _with_sinks(ws, bs):
    seen = {}
    for t in list(ws) + list(bs):
        if id(t) not in seen:
            t.requires_grad_(True)
            t._cr_grad = torch.zeros_like(t)
            seen[id(t)] = t
    return list(seen.values())


@pytest.mark.parametrize("k,shared,relu", [(3, False, False), (3, True, True), (1, False, False)])
def test_group_matches_single_convs_and_oracle_f32(k, shared, relu):
    from oracle import cpu_backend as O
    g = torch.Generator().manual_seed(5 + k)
    xs = _levels(g)
    ws, bs = _weights(g, len(xs), 128, k, shared)
    dys = [torch.randn(x.shape[0], x.shape[1], x.shape[2], 128, generator=g).to(DEV) for x in xs]
    dys[3] = None                                            # level nobody differentiates
    pad = k // 2

    def run(grouped):
        prev = ops._GROUP_ON[0]
        ops._GROUP_ON[0] = grouped
        try:
            xin = [x.clone().requires_grad_(True) for x in xs]
            params = _with_sinks(ws, bs)
            ys = ops.conv_bias_act_group(xin, ws, bs, pad=pad, relu=relu)
            outs = [y for y, d in zip(ys, dys) if d is not None]
            torch.autograd.backward(outs, [d for d in dys if d is not None])
            return [y.detach() for y in ys], [x.grad for x in xin], [p._cr_grad.clone() for p in params]
        finally:
            ops._GROUP_ON[0] = prev
            for t in list(ws) + list(bs):
                t.requires_grad_(False)
    ya, dxa, dpa = run(True)
    yb, dxb, dpb = run(False)
    for a, b in zip(ya, yb):
        assert _rel(a, b) < 2e-6
    for i, (a, b) in enumerate(zip(dxa, dxb)):
        if dys[i] is None:
            assert a is None and b is None
        else:
            assert _rel(a, b) < 2e-6
    for a, b in zip(dpa, dpb):
        assert _rel(a, b) < 2e-5
    # float32 CPU oracle (ATen) on the first two levels
    for i in range(2):
        xo = xs[i].cpu().clone().requires_grad_(True)
        wo = ws[i].detach().cpu().clone().requires_grad_(True)
        bo = bs[i].detach().cpu().clone().requires_grad_(True)
        yo = O.conv_bias_act(xo, wo, bo, 1, pad, relu=relu)
        yo.backward(dys[i].cpu())
        assert _rel(ya[i].cpu(), yo.detach()) < 2e-5 and _rel(dxa[i].cpu(), xo.grad) < 2e-5


def test_group_bf16_forward_and_input_gradient():
    g = torch.Generator().manual_seed(9)
    xs = _levels(g, dt=torch.bfloat16)
    ws, bs = _weights(g, len(xs), 128, 3, False)
    prev = ops.set_precision("bf16")
    try:
        def run(grouped):
            p = ops._GROUP_ON[0]
            ops._GROUP_ON[0] = grouped
            try:
                xin = [x.clone().requires_grad_(True) for x in xs]
                ys = ops.conv_bias_act_group(xin, ws, bs, pad=1)
                torch.autograd.backward(ys, [torch.ones_like(y) for y in ys])
                return [y.detach() for y in ys], [x.grad for x in xin]
            finally:
                ops._GROUP_ON[0] = p
        ya, dxa = run(True)
        yb, dxb = run(False)
        for a, b in zip(ya + dxa, yb + dxb):
            assert _rel(a, b) < 1e-2
    finally:
        ops.set_precision(prev)


def test_group_folds_a_gradient_slot_contribution():
    """a later consumer of a group input (the RoI pooler of the pyramid maps) leaves its gradient in the slot; the group's
    backward-data adds it in its epilogue"""
    g = torch.Generator().manual_seed(3)
    xs = [x.requires_grad_(True) for x in _levels(g, sizes=((16, 16), (8, 8)))]
    ws, bs = _weights(g, 2, 128, 3, True)
    ys = ops.conv_bias_act_group(xs, ws, bs, pad=1)
    extra = [torch.randn_like(x) for x in xs]
    for x, e in zip(xs, extra):                          # what a second registered consumer would do in its backward
        slot, idx = ops._slot_register(x, False)
        assert slot is not None and idx == 2
        ops._slot_put(slot, e)
    torch.autograd.backward(ys, [torch.ones_like(y) for y in ys])
    xs2 = [x.detach().clone().requires_grad_(True) for x in xs]
    prev = ops._GROUP_ON[0]
    ops._GROUP_ON[0] = False
    try:
        y2 = ops.conv_bias_act_group(xs2, ws, bs, pad=1)
    finally:
        ops._GROUP_ON[0] = prev
    torch.autograd.backward(y2, [torch.ones_like(y) for y in y2])
    for x, x2, e in zip(xs, xs2, extra):
        assert _rel(x.grad, x2.grad + e) < 2e-6


def test_root_conv_bn_children_gradients_equal_the_concatenation_path():
    """ops.root_conv_bn_act (DLA Root): per-child backward-data GEMMs on row slices of the transposed weights give the
    gradients autograd derives through torch.cat; a child that a convolution consumed first receives its share through that
    convolution's gradient slot (no autograd add)"""
    g = torch.Generator().manual_seed(21)
    N, H, W = 2, 24, 20
    chans = (64, 64, 32)
    kids = [torch.randn(N, H, W, c, generator=g).to(DEV) for c in chans]
    wr = (torch.randn(64, sum(chans), 1, 1, generator=g) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last)
    w3 = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last)
    gamma, beta = (torch.rand(64, generator=g) + 0.5).to(DEV), (torch.randn(64, generator=g) * 0.1).to(DEV)
    dy = torch.randn(N, H, W, 64, generator=g).to(DEV)

    def run(fused):
        prev = ops._ROOT_FUSED[0]
        ops._ROOT_FUSED[0] = fused
        try:
            ks = [k.clone().requires_grad_(True) for k in kids]
            wr_, w3_, ga, be = wr.clone().requires_grad_(True), w3.clone().requires_grad_(True), gamma.clone().requires_grad_(True), \
                beta.clone().requires_grad_(True)
            rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
            # child 1 is consumed by a convolution FIRST (like x1 -> tree2.conv1 in DLA), then by the root
            side = ops.conv_bias_act(ks[1], w3_, None, 1, 1)
            y = ops.root_conv_bn_act(ks, wr_, ga, be, rm, rv, relu=True)
            torch.autograd.backward([y, side], [dy, dy])
            return y.detach(), [k.grad for k in ks], wr_.grad, ga.grad, be.grad
        finally:
            ops._ROOT_FUSED[0] = prev
    ya, ga_, wa, gga, gba = run(True)
    yb, gb_, wb, ggb, gbb = run(False)
    assert torch.equal(ya, yb)
    for a, b in zip(ga_, gb_):
        assert _rel(a, b) < 2e-6
    assert _rel(wa, wb) < 2e-5 and _rel(gga, ggb) < 2e-5 and _rel(gba, gbb) < 2e-5


def test_group_tail_split_matches_single_convs():
    """a group whose tiles do not fill whole rounds of the 512 resident slots (the pyramid of 2 x 512 x 512 images: 682 tiles)
    runs its smaller problems split three ways along k (partial slabs + k_splitk_epilogue): same results as the single
    convolutions up to the summation order, bias / the gradient-slot contribution applied by the epilogue (no ReLU here: a
    pre-activation within rounding of zero may fall on either side of it under a different summation order)"""
    g = torch.Generator().manual_seed(77)
    sizes = ((128, 128), (64, 64), (32, 32), (16, 16), (8, 8))
    xs = _levels(g, N=2, C=256, sizes=sizes)
    ws, bs = _weights(g, len(xs), 256, 3, True)
    dys = [torch.randn(x.shape, generator=g).to(DEV) for x in xs]

    def run(grouped):
        prev = ops._GROUP_ON[0]
        ops._GROUP_ON[0] = grouped
        try:
            xin = [x.clone().requires_grad_(True) for x in xs]
            params = _with_sinks(ws, bs)
            ys = ops.conv_bias_act_group(xin, ws, bs, pad=1, relu=False)
            torch.autograd.backward(ys, dys)
            return [y.detach() for y in ys], [x.grad for x in xin], [p._cr_grad.clone() for p in params]
        finally:
            ops._GROUP_ON[0] = prev
            for t in list(ws) + list(bs):
                t.requires_grad_(False)
    ya, dxa, dpa = run(True)
    yb, dxb, dpb = run(False)
    for a, b in zip(ya, yb):
        assert _rel(a, b) < 3e-6
    for a, b in zip(dxa, dxb):
        assert _rel(a, b) < 3e-6
    for a, b in zip(dpa, dpb):
        assert _rel(a, b) < 2e-5
    relu_a = ops.conv_bias_act_group([x.clone() for x in xs], ws, bs, pad=1, relu=True)
    for a, b in zip(relu_a, yb):
        assert float((a - b.clamp(min=0)).abs().max()) < 1e-4 and float(a.min()) >= 0.0
