"""GPU parity of the conv-stack kernels in the REFERENCE'S precision (float32 activations and operands,
v_mfma_f32_16x16x4_f32) against the plain PyTorch float32 definition on the CPU (oracle/torch_ref.py).

The reference trains in float32 throughout (tools/train_net.py:184-330 of the reference, no autocast), so this is the
mode the headline benchmark runs in.  The f32 MFMA is an exact fmaf chain (one rounding per product): the only
difference to ATen's CPU convolution is the summation order, so the tolerance is 2e-5 of the tensor's scale for
activations and 1e-4 for gradients that sum over up to 65 536 pixels (stated per assert).  Includes the bench's dominant
layer (3x3 256->256 on 4x128x128) in all three directions and the large-tile / LDS-DMA variants, which VERDICT r1 found
untested at full size."""
import importlib

import pytest
import torch
import torch.nn.functional as F

from oracle import torch_ref as R

pytestmark = pytest.mark.gpu
ops = importlib.import_module("3dod_amd.hipops")
DEV = "cuda:0"
f32 = torch.float32


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def relerr(a, b):
    a = a.float().cpu(); b = b.float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def l2err(a, b):
    a = a.float().cpu(); b = b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def mk_weight(co, ci, k, g):
    return torch.randn(co, ci, k, k, generator=g) * (2.0 / (k * k * ci)) ** 0.5


CASES = [  # N, H, W, Cin, Cout, k, stride, pad
    (2, 16, 16, 8, 16, 7, 1, 3),        # stem shape class (Kdim = 392: k tail inside a 16-wide sub-step)
    (2, 16, 16, 4, 16, 7, 1, 3),        # the f32 stem as the model runs it: RGB padded to ONE 16-B chunk = 4 channels
    (2, 64, 64, 64, 128, 3, 2, 1),      # stride 2, backward-data by output-pixel parity class (4 + 2 + 2 + 1 taps)
    (2, 32, 32, 16, 16, 3, 1, 1),
    (2, 32, 32, 16, 32, 3, 2, 1),
    (1, 24, 40, 32, 64, 3, 2, 1),       # non-square, M tail
    (2, 16, 16, 64, 64, 3, 1, 1),
    (2, 16, 16, 128, 128, 3, 1, 1),
    (1, 8, 8, 256, 256, 3, 2, 1),
    (2, 16, 16, 128, 64, 1, 1, 0),      # Root 1x1
    (2, 16, 16, 320, 128, 1, 1, 0),     # Root with level_root concat (Cin not a power of two)
    (3, 10, 14, 64, 128, 1, 1, 0),      # project 1x1, odd sizes
    (2, 64, 64, 64, 128, 3, 1, 1),      # 64x64 tiles, >= 192 blocks
    (4, 64, 64, 128, 128, 3, 1, 1),     # 128x128 tiles through the LDS-DMA kernel (Cin % 32 == 0, >= 128 big tiles)
    (3, 48, 32, 16, 16, 3, 1, 1),       # stem patch weight-gradient kernel: non-square, 18 tiles
    (2, 32, 48, 4, 16, 7, 1, 3),        # ... its 7x7 / 4-channel form (four taps per accumulator tile, 49 = 12 * 4 + 1)
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("relu,use_res", [(True, False), (True, True), (False, False)])
def test_conv_bn_act_fwd_bwd_f32(case, relu, use_res):
    N, H, W, Ci, Co, k, st, pd = case
    g = torch.Generator().manual_seed(hash(case) % 2**31)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = mk_weight(Co, Ci, k, g)
    gamma = torch.rand(Co, generator=g) + 0.5
    beta = torch.randn(Co, generator=g) * 0.1
    Ho, Wo = (H + 2 * pd - k) // st + 1, (W + 2 * pd - k) // st + 1
    res = torch.randn(N, Co, Ho, Wo, generator=g) if use_res else None
    dy = torch.randn(N, Co, Ho, Wo, generator=g)
    xo, wo, go, bo = [t.clone().requires_grad_(True) for t in (x, w, gamma, beta)]
    ro = res.clone().requires_grad_(True) if use_res else None
    yo = R.conv_bn_act(xo, wo, go, bo, st, pd, relu, ro)
    yo.backward(dy)
    xd = nhwc(x).to(DEV).requires_grad_(Ci >= 16)
    wd = w.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gd, bd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    rd = nhwc(res).to(DEV).requires_grad_(True) if use_res else None
    rm, rv = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
    yd = ops.conv_bn_act(xd, wd, gd, bd, rm, rv, st, pd, relu, rd)
    assert yd.dtype == f32
    yd.backward(nhwc(dy).to(DEV))
    torch.cuda.synchronize()
    assert relerr(nchw(yd), yo.detach()) < 2e-5
    # ReLU masks agree except where |pre-activation| < ~1e-6: an L2 bound covers those isolated flips
    tol = 1e-4 if not relu else 2e-3
    err = relerr if not relu else l2err
    if Ci >= 16:
        assert err(nchw(xd.grad), xo.grad) < tol
    assert err(wd.grad, wo.grad) < tol
    assert err(gd.grad, go.grad) < tol
    assert err(bd.grad, bo.grad) < tol
    if use_res:
        assert l2err(nchw(rd.grad), ro.grad) < 2e-3
    yraw = F.conv2d(x, w, None, st, pd)
    assert relerr(rm, 0.1 * yraw.mean((0, 2, 3))) < 1e-4
    assert relerr(rv, 0.9 + 0.1 * yraw.var((0, 2, 3), unbiased=True)) < 1e-4


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 256, 1, 1, 0), (2, 16, 16, 256, 256, 3, 1, 1),
                                  (1, 8, 8, 512, 256, 1, 1, 0), (2, 16, 16, 256, 16, 1, 1, 0),
                                  (2, 32, 48, 16, 16, 3, 1, 1)])     # last: patch weight-gradient kernel with bias
@pytest.mark.parametrize("relu", [False, True])
def test_conv_bias_act_fwd_bwd_f32(case, relu):
    N, H, W, Ci, Co, k, st, pd = case
    g = torch.Generator().manual_seed(7 + Ci + Co)
    x = torch.randn(N, Ci, H, W, generator=g); w = mk_weight(Co, Ci, k, g)
    b = torch.randn(Co, generator=g) * 0.1
    xo, wo, bo = [t.clone().requires_grad_(True) for t in (x, w, b)]
    yo = R.conv_bias_act(xo, wo, bo, st, pd, relu)
    dy = torch.randn_like(yo)
    yo.backward(dy)
    xd = nhwc(x).to(DEV).requires_grad_(True)
    wd = w.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True)
    yd = ops.conv_bias_act(xd, wd, bd, st, pd, relu)
    assert yd.dtype == f32
    yd.backward(nhwc(dy).to(DEV))
    torch.cuda.synchronize()
    assert relerr(nchw(yd), yo.detach()) < 2e-5
    tol, err = (1e-4, relerr) if not relu else (2e-3, l2err)
    assert err(nchw(xd.grad), xo.grad) < tol
    assert err(wd.grad, wo.grad) < tol
    assert err(bd.grad, bo.grad) < tol


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_dominant_layer_all_directions_full_size(prec):
    """3x3 256->256 on 4x128x128 (the FPN p2 output convolution: the largest conv of the train step and the kernel the
    bench's roofline line is quoted on), forward (LDS-DMA 128x128 tiles), backward-data and the split weight gradient,
    each against the float32 definition.  fp32: 2e-5 / 1e-4.  bf16: operands rounded to bf16 first, 4e-3 of scale
    (f32 accumulation of exact bf16 products; outputs rounded to bf16 once)."""
    dt = f32 if prec == "fp32" else torch.bfloat16
    g = torch.Generator().manual_seed(11)
    N, H, W, C = 4, 128, 128, 256
    x = torch.randn(N, C, H, W, generator=g).to(dt).float()
    w = mk_weight(C, C, 3, g).to(dt).float()
    dy = torch.randn(N, C, H, W, generator=g).to(dt).float()
    xo, wo = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    yo = F.conv2d(xo, wo, None, 1, 1)
    yo.backward(dy)
    xd = nhwc(x).to(DEV).to(dt)
    w4 = w.to(DEV).contiguous(memory_format=torch.channels_last)
    wb, wt = ops.prepared_weights(w4, True, dt)
    yd = ops.conv_fwd_raw(xd, wb, C, 3, 1, 1)
    dyd = nhwc(dy).to(DEV).to(dt)
    dxd = ops.conv_bwd_data_raw(dyd, wt, xd.shape, 3, 1, 1)
    dwd = ops.conv_bwd_weight_raw(dyd, xd, 3, 1, 1)
    torch.cuda.synchronize()
    t_act, t_grad = (2e-5, 1e-4) if prec == "fp32" else (4e-3, 4e-3)
    assert relerr(nchw(yd), yo.detach()) < t_act
    assert relerr(nchw(dxd), xo.grad) < t_act
    assert relerr(dwd, wo.grad) < t_grad


def test_pool_upsample_preprocess_f32():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 12, 20, generator=g).clamp(min=0)
    for fn_d, fn_o in ((ops.maxpool2x2, lambda t: F.max_pool2d(t, 2, 2)), (ops.subsample2x, lambda t: F.max_pool2d(t, 1, 2)),
                       (ops.maxpool3x3s2, lambda t: F.max_pool2d(t, 3, 2, 1))):
        xo = x.clone().requires_grad_(True)
        yo = fn_o(xo); dy = torch.randn_like(yo); yo.backward(dy)
        xd = nhwc(x).to(DEV).requires_grad_(True)
        yd = fn_d(xd); yd.backward(nhwc(dy).to(DEV))
        assert yd.dtype == f32
        assert torch.equal(nchw(yd).cpu(), yo.detach())
        assert torch.allclose(nchw(xd.grad).cpu(), xo.grad, rtol=1e-6, atol=1e-6)
    lat, top = torch.randn(2, 16, 8, 12, generator=g), torch.randn(2, 16, 4, 6, generator=g)
    lo, to = lat.clone().requires_grad_(True), top.clone().requires_grad_(True)
    yo = R.upsample2x_add(lo, to); dy = torch.randn_like(yo); yo.backward(dy)
    ld, td = [nhwc(t).to(DEV).requires_grad_(True) for t in (lat, top)]
    yd = ops.upsample2x_add(ld, td); yd.backward(nhwc(dy).to(DEV))
    assert torch.equal(nchw(yd).cpu(), yo.detach())
    assert relerr(nchw(ld.grad), lo.grad) < 1e-6 and relerr(nchw(td.grad), to.grad) < 1e-6
    img = torch.randint(0, 256, (2, 3, 8, 12), generator=g, dtype=torch.uint8)
    mean, std = [103.530, 116.280, 123.675], [57.375, 57.120, 58.395]
    y = ops.preprocess(img.to(DEV), mean, std, dtype=f32).cpu()
    ref = (img.float() - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    assert relerr(y[..., :3], nhwc(ref)) < 1e-6 and (y[..., 3:] == 0).all()


@pytest.mark.parametrize("C", [16, 256])       # 256: a block per (RoI, 32-channel group = XCD) in the backward
def test_roi_align_fwd_bwd_f32(C):
    g = torch.Generator().manual_seed(5)
    N = 2
    sizes = [(32, 40), (16, 20), (8, 10), (4, 5), (2, 3)]
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32, 1 / 64]
    feats = [torch.randn(N, C, h, w, generator=g) for h, w in sizes]
    wh = torch.tensor([[20., 24.], [60, 50], [110, 130], [300, 200], [700, 600], [15, 90], [40, 40], [128, 160]])
    ctr = torch.rand(8, 2, generator=g) * torch.tensor([160., 128.])
    rois = torch.cat([torch.tensor([[0.], [1], [0], [1], [0], [1], [0], [1]]), ctr - wh / 2, ctr + wh / 2], 1)
    fo = [f.clone().requires_grad_(True) for f in feats]
    yo = R.roi_align(fo, rois, scales, 7)
    dy = torch.randn_like(yo); yo.backward(dy)
    fd = [nhwc(f).to(DEV).requires_grad_(True) for f in feats]
    yd = ops.roi_align_pyramid(fd, rois.to(DEV), scales, 7)
    assert yd.dtype == f32
    yd.backward(nhwc(dy).to(DEV))
    torch.cuda.synchronize()
    assert relerr(nchw(yd), yo.detach()) < 1e-5
    for a, b in zip(fd, fo):
        if b.grad is None:
            assert float(a.grad.abs().max()) == 0.0
        else:
            assert relerr(nchw(a.grad), b.grad) < 1e-4


@pytest.mark.parametrize("chw", [None, (32, 7, 7), (80, 3, 5)])
def test_linear_fc_f32(chw):
    """ops.linear in f32: the implicit-GEMM kernels as a plain GEMM (1x1 convolution over a (1,1,rows,K) map), with the
    (c,h,w)->(h,w,c) column re-ordering of the first RoI-head FC and gradients into the parameters' sinks."""
    from oracle import cpu_backend as O
    g = torch.Generator().manual_seed(17)
    K = 96 if chw is None else chw[0] * chw[1] * chw[2]
    n, Odim = 50, 48
    x, w, b = torch.randn(n, K, generator=g), torch.randn(Odim, K, generator=g) * 0.1, torch.randn(Odim, generator=g)
    dy = torch.randn(n, Odim, generator=g)
    xo, wo, bo = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yo = O.linear(xo, wo, bo, chw=chw)
    yo.backward(dy)
    xd, wd, bd = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    yd = ops.linear(xd, wd, bd, chw=chw)
    assert yd.dtype == f32
    yd.backward(dy.to(DEV))
    assert relerr(yd.cpu(), yo.detach()) < 2e-5
    assert relerr(xd.grad.cpu(), xo.grad) < 2e-5
    assert relerr(wd.grad.cpu(), wo.grad) < 1e-4 and relerr(bd.grad.cpu(), bo.grad) < 1e-4
    wd2, bd2 = w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    wd2._cr_grad, bd2._cr_grad = torch.ones_like(wd2), torch.ones_like(bd2)
    ops.linear(xd.detach(), wd2, bd2, chw=chw).backward(dy.to(DEV))
    assert wd2.grad is None and bd2.grad is None
    assert relerr((wd2._cr_grad - 1).cpu(), wo.grad) < 1e-4 and relerr((bd2._cr_grad - 1).cpu(), bo.grad) < 1e-4


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_linear_cat_predictors(prec):
    """the predictor layers of a head as one GEMM with rows padded to the MFMA tile (51 + 200 -> 256 columns): values,
    and gradients routed back to every parameter"""
    dt = f32 if prec == "fp32" else torch.bfloat16
    g = torch.Generator().manual_seed(19)
    n, K = 70, 128
    x = torch.randn(n, K, generator=g).to(dt).float()
    ws = [(torch.randn(o, K, generator=g) * 0.1).to(dt).float() for o in (51, 200)]
    bs = [torch.randn(o, generator=g).to(dt).float() for o in (51, 200)]
    xo = x.clone().requires_grad_()
    wo, bo = [w.clone().requires_grad_() for w in ws], [b.clone().requires_grad_() for b in bs]
    yo = [F.linear(xo, w, b) for w, b in zip(wo, bo)]
    dys = [torch.randn_like(y) for y in yo]
    sum((y * d).sum() for y, d in zip(yo, dys)).backward()
    xd = x.to(DEV).to(dt).requires_grad_()
    wd, bd = [w.to(DEV).requires_grad_() for w in ws], [b.to(DEV).requires_grad_() for b in bs]
    y, offs = ops.linear_cat(xd, wd, bd)
    assert y.shape == (n, 256) and offs == [0, 51, 251]
    sum((y[:, offs[i]:offs[i + 1]] * dys[i].to(DEV)).sum() for i in range(2)).backward()
    tol = 1e-4 if prec == "fp32" else 2e-2
    for i in range(2):
        assert relerr(y[:, offs[i]:offs[i + 1]].detach().cpu(), yo[i].detach()) < tol
        assert relerr(wd[i].grad.cpu(), wo[i].grad) < tol and relerr(bd[i].grad.cpu(), bo[i].grad) < tol
    assert relerr(xd.grad.float().cpu(), xo.grad) < tol
    assert float(y[:, 251:].abs().max()) == 0.0


def test_gradient_slots_equal_autograd_adds():
    """fan-in of gradients through the backward-data epilogue (hipops._GradSlot): a residual block (x feeds conv1 and the
    residual input of conv2) and an FPN-style top-down step (prev feeds a 3x3 output conv and the next level's upsample)
    give the same input gradients with the slots on (no add kernels) and off (autograd's adds)"""
    g = torch.Generator().manual_seed(21)
    N, H, W, C = 2, 16, 16, 64
    x0 = torch.randn(N, H, W, C, generator=g).to(DEV)
    lat0 = torch.randn(N, 2 * H, 2 * W, C, generator=g).to(DEV)
    w1, w2, w3 = [(torch.randn(C, C, 3, 3, generator=g) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
                  for _ in range(3)]
    gam, bet = torch.ones(C, device=DEV, requires_grad=True), torch.zeros(C, device=DEV, requires_grad=True)
    bias = torch.zeros(C, device=DEV, requires_grad=True)

    def run(on):
        prev = ops._SLOTS_ON[0]
        ops._SLOTS_ON[0] = on
        try:
            x = x0.clone().requires_grad_(True)
            lat = lat0.clone().requires_grad_(True)
            rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
            xin = x * 1.0                                      # a non-leaf activation, as inside a network
            out = ops.conv_bn_act(xin, w1, gam, bet, rm, rv, 1, 1, True)
            blk = ops.conv_bn_act(out, w2, gam, bet, rm.clone(), rv.clone(), 1, 1, True, residual=xin)
            p = ops.conv_bias_act(blk, w3, bias, 1, 1)          # "output conv" on blk ...
            up = ops.upsample2x_add(lat, blk)                   # ... and the finer level's top-down step on the same blk
            ((p ** 2).mean() + (up ** 2).mean()).backward()
            torch.cuda.synchronize()
            return x.grad.clone(), lat.grad.clone()
        finally:
            ops._SLOTS_ON[0] = prev
    gx1, gl1 = run(True)
    gx0, gl0 = run(False)
    assert torch.equal(gl1, gl0)
    assert relerr(gx1, gx0) < 1e-6


def test_linear_head_fc1_full_size():
    """the box head's first FC layer at the train step's size (2048 RoIs x 12544 -> 1024): the forward takes the long-k
    split (128 x 128 tiles, 4 k ranges + epilogue), the weight gradient 64-row tiles with two row splits (1568 tiles);
    against torch's f32 GEMM (same arithmetic up to summation order)."""
    g = torch.Generator(device=DEV).manual_seed(3)
    R, K, O = 2048, 12544, 1024
    x = torch.randn(R, K, device=DEV, generator=g)
    w = torch.randn(O, K, device=DEV, generator=g) * 0.01
    b = torch.randn(O, device=DEV, generator=g) * 0.1
    dy = torch.randn(R, O, device=DEV, generator=g)
    y = ops.linear_fwd_raw(x, w, b, relu=True)
    ref = torch.relu(x @ w.t() + b)
    assert relerr(y, ref) < 2e-5
    dx = ops.linear_bwd_data_raw(dy, w.t().contiguous())
    assert relerr(dx, dy @ w) < 2e-5
    dw = torch.zeros(O, K, device=DEV); db = torch.zeros(O, device=DEV)
    ops.linear_bwd_weight_raw(dy, x, dw, db, True)
    assert relerr(dw, dy.t() @ x) < 2e-5
    assert relerr(db, dy.sum(0)) < 2e-5
