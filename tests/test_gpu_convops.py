"""GPU parity of the conv-stack kernels (through the C-ABI + autograd glue) against the plain
PyTorch float32 CPU oracle (oracle/torch_ref.py).  bf16 storage / f32 accumulate: tolerance
2e-2 relative to the tensor's scale for activations and gradients (stated per assert)."""
import importlib

import numpy as np
import pytest
import torch

from oracle import torch_ref as R

pytestmark = pytest.mark.gpu
ops = importlib.import_module("3dod_amd.hipops")
DEV = "cuda:0"
bf16 = torch.bfloat16


def q(t):
    """round to bf16 and back (the oracle sees exactly the values the kernel sees)."""
    return t.to(bf16).to(torch.float32)


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
    w = torch.randn(co, ci, k, k, generator=g) * (2.0 / (k * k * ci)) ** 0.5
    return q(w)


CASES = [  # N, H, W, Cin, Cout, k, stride, pad
    (2, 16, 16, 8, 16, 7, 1, 3),        # stem shape class (Kdim = 392, not a multiple of 32)
    (2, 32, 32, 16, 16, 3, 1, 1),
    (2, 32, 32, 16, 32, 3, 2, 1),
    (1, 24, 40, 32, 64, 3, 2, 1),       # non-square, M tail
    (2, 16, 16, 64, 64, 3, 1, 1),
    (2, 16, 16, 128, 128, 3, 1, 1),     # BN = 128 path
    (1, 8, 8, 256, 256, 3, 2, 1),
    (2, 16, 16, 128, 64, 1, 1, 0),      # Root 1x1
    (2, 16, 16, 320, 128, 1, 1, 0),     # Root with level_root concat (Cin not a power of two)
    (3, 10, 14, 64, 128, 1, 1, 0),      # project 1x1, odd sizes
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("relu,use_res", [(True, False), (True, True), (False, False)])
def test_conv_bn_act_fwd_bwd(case, relu, use_res):
    N, H, W, Ci, Co, k, st, pd = case
    g = torch.Generator().manual_seed(hash(case) % 2**31)
    x = q(torch.randn(N, Ci, H, W, generator=g))
    w = mk_weight(Co, Ci, k, g)
    gamma = torch.rand(Co, generator=g) + 0.5
    beta = torch.randn(Co, generator=g) * 0.1
    Ho, Wo = (H + 2 * pd - k) // st + 1, (W + 2 * pd - k) // st + 1
    res = q(torch.randn(N, Co, Ho, Wo, generator=g)) if use_res else None
    dy = q(torch.randn(N, Co, Ho, Wo, generator=g))
    # oracle
    xo, wo, go, bo = [t.clone().requires_grad_(True) for t in (x, w, gamma, beta)]
    ro = res.clone().requires_grad_(True) if use_res else None
    yo = R.conv_bn_act(xo, wo, go, bo, st, pd, relu, ro)
    yo.backward(dy)
    # HIP
    xd = nhwc(x).to(DEV).to(bf16).requires_grad_(Ci >= 16)     # the stem (Cin=8) never needs d/dx
    wd = w.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gd, bd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    rd = nhwc(res).to(DEV).to(bf16).requires_grad_(True) if use_res else None
    rm, rv = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
    yd = ops.conv_bn_act(xd, wd, gd, bd, rm, rv, st, pd, relu, rd)
    yd.backward(nhwc(dy).to(DEV).to(bf16))
    torch.cuda.synchronize()
    assert relerr(nchw(yd), yo.detach()) < 2e-2
    # With ReLU, activations stored in bf16 flip the mask of the few elements whose pre-activation is
    # within bf16 rounding of zero; each flip is an O(|dy|) difference at one element.  So: exact (max-norm)
    # check of the whole backward without ReLU, and with ReLU an elementwise check of the masked gradient
    # away from zero plus an L2 bound on the rest.
    if not relu:
        if Ci >= 16:
            assert relerr(nchw(xd.grad), xo.grad) < 3e-2
        assert relerr(wd.grad, wo.grad) < 3e-2
        assert relerr(gd.grad, go.grad) < 3e-2
        assert relerr(bd.grad, bo.grad) < 3e-2
    else:
        if Ci >= 16:
            assert l2err(nchw(xd.grad), xo.grad) < 0.15
        assert l2err(wd.grad, wo.grad) < 0.15
        assert l2err(gd.grad, go.grad) < 0.1
        assert l2err(bd.grad, bo.grad) < 0.1
    if use_res:
        with torch.no_grad():
            z = torch.nn.functional.batch_norm(torch.nn.functional.conv2d(x, w, None, st, pd), None, None, gamma, beta,
                                               True, 0.1, 1e-5) + res
        safe = z.abs() > 0.05
        diff = (nchw(rd.grad).float().cpu() - ro.grad).abs()
        assert float(diff[safe].max()) < 2e-2 * float(ro.grad.abs().max())
        assert l2err(nchw(rd.grad), ro.grad) < 0.15
    # running statistics follow nn.BatchNorm2d (momentum 0.1, unbiased variance)
    yraw = torch.nn.functional.conv2d(x, w, None, st, pd)
    assert relerr(rm, 0.1 * yraw.mean((0, 2, 3))) < 2e-2
    assert relerr(rv, 0.9 + 0.1 * yraw.var((0, 2, 3), unbiased=True)) < 2e-2


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 256, 1, 1, 0), (2, 16, 16, 256, 256, 3, 1, 1),
                                  (1, 8, 8, 512, 256, 1, 1, 0), (2, 16, 16, 256, 16, 1, 1, 0)])
@pytest.mark.parametrize("relu,out_f32", [(False, False), (True, False), (False, True)])
def test_conv_bias_act_fwd_bwd(case, relu, out_f32):
    N, H, W, Ci, Co, k, st, pd = case
    g = torch.Generator().manual_seed(7 + Ci + Co)
    x = q(torch.randn(N, Ci, H, W, generator=g)); w = mk_weight(Co, Ci, k, g)
    b = torch.randn(Co, generator=g) * 0.1
    xo, wo, bo = [t.clone().requires_grad_(True) for t in (x, w, b)]
    yo = R.conv_bias_act(xo, wo, bo, st, pd, relu)
    dy = q(torch.randn_like(yo))
    yo.backward(dy)
    xd = nhwc(x).to(DEV).to(bf16).requires_grad_(True)
    wd = w.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True)
    yd = ops.conv_bias_act(xd, wd, bd, st, pd, relu, out_f32)
    assert yd.dtype == (torch.float32 if out_f32 else bf16)
    yd.backward(nhwc(dy).to(DEV).to(yd.dtype))
    torch.cuda.synchronize()
    assert relerr(nchw(yd), yo.detach()) < (2e-3 if out_f32 else 2e-2)
    assert relerr(nchw(xd.grad), xo.grad) < 3e-2
    assert relerr(wd.grad, wo.grad) < 3e-2
    assert relerr(bd.grad, bo.grad) < 3e-2


def test_pool_and_fpn_topdown():
    g = torch.Generator().manual_seed(3)
    x = q(torch.randn(2, 32, 12, 20, generator=g)).clamp(min=0)      # post-ReLU like: many exact ties at 0
    for fn_d, fn_o in ((ops.maxpool2x2, lambda t: torch.nn.functional.max_pool2d(t, 2, 2)),
                       (ops.subsample2x, lambda t: torch.nn.functional.max_pool2d(t, 1, 2))):
        xo = x.clone().requires_grad_(True)
        yo = fn_o(xo); dy = q(torch.randn_like(yo)); yo.backward(dy)
        xd = nhwc(x).to(DEV).to(bf16).requires_grad_(True)
        yd = fn_d(xd); yd.backward(nhwc(dy).to(DEV).to(bf16))
        assert torch.equal(nchw(yd).float().cpu(), yo.detach())
        assert torch.equal(nchw(xd.grad).float().cpu(), xo.grad)           # first-max tie rule
    lat = q(torch.randn(2, 16, 8, 12, generator=g)); top = q(torch.randn(2, 16, 4, 6, generator=g))
    lo, to = lat.clone().requires_grad_(True), top.clone().requires_grad_(True)
    yo = R.upsample2x_add(lo, to); dy = q(torch.randn_like(yo)); yo.backward(dy)
    ld, td = [nhwc(t).to(DEV).to(bf16).requires_grad_(True) for t in (lat, top)]
    yd = ops.upsample2x_add(ld, td); yd.backward(nhwc(dy).to(DEV).to(bf16))
    assert relerr(nchw(yd), yo.detach()) < 1e-2
    assert relerr(nchw(ld.grad), lo.grad) < 1e-6 and relerr(nchw(td.grad), to.grad) < 1e-2


def test_preprocess():
    g = torch.Generator().manual_seed(4)
    img = torch.randint(0, 256, (2, 3, 8, 12), generator=g, dtype=torch.uint8)
    mean, std = [103.530, 116.280, 123.675], [57.375, 57.120, 58.395]
    y = ops.preprocess(img.to(DEV), mean, std).float().cpu()
    ref = (img.float() - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    assert relerr(y[..., :3], nhwc(ref)) < 1e-2
    assert (y[..., 3:] == 0).all()


def test_roi_align_fwd_bwd():
    g = torch.Generator().manual_seed(5)
    C, N = 16, 2
    sizes = [(32, 40), (16, 20), (8, 10), (4, 5), (2, 3)]
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32, 1 / 64]
    feats = [q(torch.randn(N, C, h, w, generator=g)) for h, w in sizes]
    # rois spanning all levels, some partially outside the image
    wh = torch.tensor([[20., 24.], [60, 50], [110, 130], [300, 200], [700, 600], [15, 90], [40, 40], [128, 160]])
    ctr = torch.rand(8, 2, generator=g) * torch.tensor([160., 128.])
    rois = torch.cat([torch.tensor([[0.], [1], [0], [1], [0], [1], [0], [1]]), ctr - wh / 2, ctr + wh / 2], 1)
    fo = [f.clone().requires_grad_(True) for f in feats]
    yo = R.roi_align(fo, rois, scales, 7)
    dy = q(torch.randn_like(yo)); yo.backward(dy)
    fd = [nhwc(f).to(DEV).to(bf16).requires_grad_(True) for f in feats]
    yd = ops.roi_align_pyramid(fd, rois.to(DEV), scales, 7)
    yd.backward(nhwc(dy).to(DEV).to(bf16))
    torch.cuda.synchronize()
    assert relerr(nchw(yd), yo.detach()) < 1e-2
    for a, b in zip(fd, fo):
        if b.grad is None:
            assert float(a.grad.abs().max()) == 0.0
        else:
            assert relerr(nchw(a.grad), b.grad) < 2e-2


@pytest.mark.parametrize("maxn,counts", [(300, [300, 257, 64, 1, 0]), (2000, [2000, 1999, 1025, 640, 63]),
                                         (2048, [2048, 2047, 1, 0, 130]), (2300, [2300, 2049, 64, 1, 0])])
def test_nms_grouped(maxn, counts):
    """maxn <= 2048 takes the register-pipelined walk (k_nms_scan32), larger groups the generic one."""
    g = torch.Generator().manual_seed(6)
    G = len(counts)
    counts = torch.tensor(counts, dtype=torch.int32)
    ctr = torch.rand(G, maxn, 2, generator=g) * 200; wh = torch.rand(G, maxn, 2, generator=g) * 60 + 5
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 2)
    boxes[0, 10] = boxes[0, 3]          # exact duplicate -> suppressed
    keep = ops.nms_grouped(boxes.to(DEV), counts.to(DEV), 0.5).cpu()
    for gi in range(G):
        n = int(counts[gi])
        scores = torch.arange(n, 0, -1, dtype=torch.float32)      # already sorted by descending score
        ref = torch.zeros(maxn, dtype=torch.bool)
        if n:
            ref[R.nms(boxes[gi, :n], scores, 0.5)] = True
        assert torch.equal(keep[gi], ref), gi


def test_sgd_and_nonfinite():
    g = torch.Generator().manual_seed(8)
    n = 100003
    p = torch.randn(n, generator=g); gr = torch.randn(n, generator=g); m = torch.randn(n, generator=g)
    pn, mn = R.sgd_step(p, gr, m, 0.02, 0.9, 1e-4)
    pd, gd, md = p.to(DEV), gr.to(DEV), m.to(DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.nonfinite_flag(gd, flag)
    ops.sgd_step(pd, gd, md, 0.02, 0.9, 1e-4, 1.0, flag)
    assert int(flag.item()) == 0
    np.testing.assert_allclose(pd.cpu().numpy(), pn.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(md.cpu().numpy(), mn.numpy(), rtol=1e-6, atol=1e-7)
    gd[777] = float("nan")
    ops.nonfinite_flag(gd, flag)
    before = pd.clone()
    ops.sgd_step(pd, gd, md, 0.02, 0.9, 1e-4, 1.0, flag)       # skipped on device
    assert int(flag.item()) == 1 and torch.equal(pd, before)


def test_weight_bank_matches_per_tensor_prep():
    """cr_weights_prepare (one launch for all conv weights) == cr_cast_f32_to_bf16 + cr_weight_transpose per tensor,
    bit for bit, including ragged 32x32 tiles (Cin/Cout not multiples of 32)."""
    g = torch.Generator().manual_seed(31)
    shapes = [(128, 64, 3), (32, 48, 1), (16, 8, 7), (40, 24, 3), (256, 256, 1)]
    sizes = [co * ci * k * k for co, ci, k in shapes]
    flat = torch.randn(sum((n + 3) // 4 * 4 for n in sizes) + 8, generator=g).to(DEV)
    params, off = [], 4
    for (co, ci, k), n in zip(shapes, sizes):
        v = flat[off:off + n].view(co, k, k, ci).permute(0, 3, 1, 2)          # logical KCRS over physical KRSC
        params.append(torch.nn.Parameter(v))
        params[-1].data = v
        off += (n + 3) // 4 * 4
    ref = []
    for p in params:
        wb, wt = ops.prepared_weights(p, True)
        ref.append((wb.clone(), wt.clone()))
    bank = ops.WeightBank(params, flat, dtype=bf16)
    ops.bump_weight_epoch()
    for i, p in enumerate(params):
        wb, wt = ops.prepared_weights(p, True)
        assert wb.data_ptr() == bank.views[i][0].data_ptr()
        assert torch.equal(wb.view(torch.int16), ref[i][0].view(torch.int16)), i
        assert torch.equal(wt.view(torch.int16), ref[i][1].view(torch.int16)), i
    flat.mul_(2.0)                                   # "optimizer step": raw update + epoch bump -> one refresh
    ops.bump_weight_epoch()
    wb, _ = ops.prepared_weights(params[0], False)
    assert torch.equal(wb.float(), (ref[0][0].float() * 2.0))
    # float32 bank (reference precision): the master weights are the forward operand, only the bwd-data layout is a copy
    bank32 = ops.WeightBank(params, flat, dtype=torch.float32)
    ops.bump_weight_epoch()
    for i, p in enumerate(params):
        wb, wt = ops.prepared_weights(p, True, torch.float32)
        assert wb.data_ptr() == p.data_ptr() and wt.data_ptr() == bank32.views[i][1].data_ptr()
        co, ci, k, _ = p.shape
        assert torch.equal(wt.view(ci, k * k, co), p.detach().permute(1, 2, 3, 0).reshape(ci, k * k, co)), i


@pytest.mark.parametrize("hw", [(12, 20), (13, 21), (2, 2)])
def test_maxpool3x3s2(hw):
    """ResNet stem pool: forward values, backward routing (overlapping windows accumulate, first-max ties) == ATen;
    the bf16 sum of up to 4 window gradients is compared after the same rounding."""
    g = torch.Generator().manual_seed(13)
    x = q(torch.randn(2, 16, *hw, generator=g)).clamp(min=0)          # post-ReLU like: many exact ties at 0
    xo = x.clone().requires_grad_(True)
    yo = torch.nn.functional.max_pool2d(xo, 3, 2, 1)
    dy = q(torch.randn_like(yo)); yo.backward(dy)
    xd = nhwc(x).to(DEV).to(bf16).requires_grad_(True)
    yd = ops.maxpool3x3s2(xd); yd.backward(nhwc(dy).to(DEV).to(bf16))
    assert torch.equal(nchw(yd).float().cpu(), yo.detach())
    assert torch.allclose(nchw(xd.grad).float().cpu(), xo.grad, rtol=1e-2, atol=1e-2)
    assert torch.equal(nchw(xd.grad).float().cpu() != 0, xo.grad != 0)       # same routing


@pytest.mark.parametrize("chw", [None, (32, 7, 7), (80, 3, 5)])
def test_linear_fc(chw):
    """ops.linear in the bf16 mode: cached bf16 weight copy (with the (c,h,w)->(h,w,c) column re-ordering of the first
    RoI-head FC), the implicit-GEMM kernels as a plain GEMM (cr_linear_*), weight / bias gradients from the GEMM's own
    epilogue -- against F.linear on the re-ordered weight."""
    from oracle import cpu_backend as O
    g = torch.Generator().manual_seed(17)
    K = 96 if chw is None else chw[0] * chw[1] * chw[2]
    n, Odim = 50, 48
    x, w, b = q(torch.randn(n, K, generator=g)), q(torch.randn(Odim, K, generator=g) * 0.1), q(torch.randn(Odim, generator=g))
    dy = q(torch.randn(n, Odim, generator=g))
    xo, wo, bo = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yo = O.linear(xo, wo, bo, chw=chw)
    yo.backward(dy)
    xd, wd, bd = x.to(DEV).to(bf16).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    yd = ops.linear(xd, wd, bd, chw=chw)
    yd.backward(dy.to(DEV).to(bf16))
    assert relerr(yd.float().cpu(), yo.detach()) < 1e-2
    assert relerr(xd.grad.float().cpu(), xo.grad) < 1e-2
    assert relerr(wd.grad.cpu(), wo.grad) < 1e-2 and relerr(bd.grad.cpu(), bo.grad) < 1e-2
    # with gradient sinks (the optimizer's flat gradient): accumulated in place, autograd returns None
    wd2, bd2 = w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    wd2._cr_grad, bd2._cr_grad = torch.ones_like(wd2), torch.ones_like(bd2)
    ops.linear(xd.detach(), wd2, bd2, chw=chw).backward(dy.to(DEV).to(bf16))
    assert wd2.grad is None and bd2.grad is None
    assert relerr((wd2._cr_grad - 1).cpu(), wo.grad) < 1e-2 and relerr((bd2._cr_grad - 1).cpu(), bo.grad) < 1e-2
    # the cached copy follows the weight epoch
    with torch.no_grad():
        wd2.mul_(2.0)
    ops.bump_weight_epoch()
    y2 = ops.linear(xd.detach(), wd2, None, chw=chw)
    assert relerr(y2.float().cpu(), (yo.detach() - b) * 2) < 2e-2


def test_folded_batchnorm_inference_matches_unfolded():
    """eval-mode conv+BN(+residual, ReLU): the folded single-convolution path (no_grad) against the affine-epilogue path
    (grad enabled) and against the float32 definition"""
    import torch.nn.functional as F
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 24, 20, 64, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(128, 64, 3, 3, generator=g) * 0.05).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    gamma, beta = (torch.rand(128, generator=g) + 0.5).to(dev).requires_grad_(), (torch.randn(128, generator=g) * 0.2).to(dev).requires_grad_()
    mean, var = (torch.randn(128, generator=g) * 0.3).to(dev), (torch.rand(128, generator=g) + 0.2).to(dev)
    res = torch.randn(2, 12, 10, 128, generator=g).to(torch.bfloat16).to(dev)
    with torch.no_grad():
        folded = ops.conv_bn_act(x, w, gamma, beta, mean, var, stride=2, pad=1, relu=True, residual=res, training=False)
    unfolded = ops.conv_bn_act(x, w, gamma, beta, mean, var, stride=2, pad=1, relu=True, residual=res, training=False)
    assert unfolded.requires_grad and not folded.requires_grad
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w.detach().to(torch.bfloat16).float(), None, 2, 1)
    ref = F.relu(F.batch_norm(y, mean, var, gamma.detach(), beta.detach(), False, 0.0, 1e-5) + res.float().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    for out in (folded, unfolded.detach()):
        assert float((out.float() - ref).norm() / ref.norm()) < 1e-2
    assert float((folded.float() - unfolded.detach().float()).norm() / ref.norm()) < 1e-2


def test_folded_batchnorm_cache_follows_weights_and_statistics():
    """the folded copies are cached on the weight tensor between eval calls; an in-place weight update, a train-mode forward
    (running statistics move through raw pointers) and a weight-epoch bump each invalidate them"""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 16, 16, 64, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    gamma, beta = (torch.rand(64, generator=g) + 0.5).to(dev).requires_grad_(), torch.zeros(64, device=dev).requires_grad_()
    mean, var = torch.zeros(64, device=dev), torch.ones(64, device=dev)

    def ev():
        with torch.no_grad():
            return ops.conv_bn_act(x, w, gamma, beta, mean, var, stride=1, pad=1, relu=False, training=False).float()

    def fresh():                                           # the same computation with nothing cached
        with torch.no_grad():
            return ops.conv_bn_act(x, w.detach().clone(memory_format=torch.preserve_format).requires_grad_(), gamma, beta, mean,
                                   var, stride=1, pad=1, relu=False, training=False).float()
    y0 = ev()
    ent = w._cr_fold
    assert torch.equal(ev(), y0) and w._cr_fold is ent               # second call: cache hit
    with torch.no_grad():
        w.mul_(2.0)                                                   # version counter moves
    y1 = ev()
    assert w._cr_fold is not ent and torch.equal(y1, fresh()) and not torch.equal(y1, y0)
    ent = w._cr_fold
    ops.conv_bn_act(x * 3 + 1, w, gamma, beta, mean, var, stride=1, pad=1, relu=False, training=True)   # updates mean / var
    y2 = ev()
    assert w._cr_fold is not ent and torch.equal(y2, fresh()) and not torch.equal(y2, y1)
    ent = w._cr_fold
    ops.bump_weight_epoch()
    assert torch.equal(ev(), y2) and w._cr_fold is not ent


DMA_CASES = [  # N, H, W, Cin, Cout, k, stride  -- all large enough for the LDS-DMA kernel (>= 128 tiles of 128 pixels)
    (2, 128, 128, 64, 64, 3, 1), (1, 128, 128, 64, 128, 3, 1), (2, 96, 100, 128, 256, 1, 1), (2, 256, 256, 64, 64, 3, 2),
    (1, 131, 127, 64, 256, 3, 1)]


@pytest.mark.parametrize("case", DMA_CASES)
def test_lds_dma_conv_equals_register_staged_kernel(case):
    """k_conv_igemm_dma (bf16 output, large layers) against k_conv_igemm on the same layer: the f32-output form never takes
    the DMA kernel, accumulates in the same order and shares the epilogue, so its result rounded to bf16 must be identical
    -- with bias, residual, ReLU and the BatchNorm statistics rows.  Backward-data against the float32 definition."""
    import torch.nn.functional as F
    N, H, W, Cin, Cout, k, stride = case
    pad = k // 2
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16).to(dev)
    w4 = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).to(dev).contiguous(memory_format=torch.channels_last)
    wb, wt = ops.prepared_weights(w4, need_transposed=True)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    assert (N * Ho * Wo + 127) // 128 * max(Cout // 128, 1) >= 128
    # (more than 320 tiles of 64 x 64: the register-staged path then has no intra-block split-K, whose sum order differs)
    assert (N * Ho * Wo + 63) // 64 * (Cout // 64) > 320
    bias = torch.randn(Cout, generator=g).to(dev)
    res = torch.randn(N, Ho, Wo, Cout, generator=g).to(torch.bfloat16).to(dev)
    nparts = (N * Ho * Wo + 63) // 64
    for kw in (dict(), dict(bias=bias, relu=True), dict(residual=res, relu=True)):
        a = ops.conv_fwd_raw(x, wb, Cout, k, stride, pad, **kw)
        b = ops.conv_fwd_raw(x, wb, Cout, k, stride, pad, out_f32=True, **kw)
        assert torch.equal(a, b.to(torch.bfloat16)), kw.keys()
    sa, sb = (torch.empty((nparts, 2, Cout), device=dev) for _ in range(2))
    a = ops.conv_fwd_raw(x, wb, Cout, k, stride, pad, stats=sa)
    b = ops.conv_fwd_raw(x, wb, Cout, k, stride, pad, stats=sb, out_f32=True)
    assert torch.equal(a, b.to(torch.bfloat16))
    # statistics rows are per 64 pixels; a 128-pixel tile puts its sums in the even row (zeros in the odd one), so compare totals
    assert torch.allclose(sa.sum(0), sb.sum(0), rtol=1e-5, atol=1e-3)
    # backward-data (also the DMA kernel: M = N*H*W pixels of dX)
    dy = torch.randn(N, Ho, Wo, Cout, generator=g).to(torch.bfloat16).to(dev)
    dx = ops.conv_bwd_data_raw(dy, wt, x.shape, k, stride, pad).float()
    xr = torch.zeros(N, Cin, H, W, device=dev, requires_grad=True)
    y = F.conv2d(xr, w4.detach().to(torch.bfloat16).float(), None, stride, pad)
    y.backward(dy.float().permute(0, 3, 1, 2))
    ref = xr.grad.permute(0, 2, 3, 1)
    assert float((dx - ref).norm() / ref.norm()) < 6e-3
