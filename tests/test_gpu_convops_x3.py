"""The split mode (act_f32 = 2, hipops precision "fp32x3"): float32 tensors, contractions on the bf16 matrix cores with
every operand split EXACTLY into three bf16 values and six of the nine cross products kept (include/cr3dod.h).

 * the split is exact and the plane layout is the documented one (cr_weight_split3, bit-level check);
 * forward, backward-data and weight gradient of the layer shapes the split kernels take (3x3 / 1x1, 128- and 64-wide
   tiles, split-K, M tails, FC-shaped GEMMs) against a float64 definition, side by side with the f32-MFMA mode: the
   split mode's error must stay within 2x the f32 MFMA's (+3e-7 of scale) -- measured: 0.8x ... 1.6x, both are
   summation-order noise of float32 accumulation (5e-7 ... 1.3e-6 of scale at k = 1024 ... 2304);
 * layers the split kernels do not cover fall back to the f32 MFMA kernels (same results as precision "fp32").
The reference's arithmetic is float32 (tools/train_net.py:184-330 of the reference); this mode reproduces it to the same
accuracy as the f32 MFMA path."""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ops = importlib.import_module("3dod_amd.hipops")
_lib = importlib.import_module("3dod_amd._lib")
DEV = "cuda:0"
f32, f64, bf16 = torch.float32, torch.float64, torch.bfloat16


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def relerr(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def test_weight_split3_exact_and_layout():
    g = torch.Generator().manual_seed(5)
    rows, K = 48, 96
    w = torch.randn(rows, K, generator=g) * torch.exp(torch.randn(rows, K, generator=g) * 8.0)      # 1e-15 .. 1e15
    w[0, :4] = torch.tensor([0.0, -0.0, 1.0, -1.0])
    w[1, :4] = torch.tensor([3.0e-30, 1.1754944e-38, 2.0 ** 100, -(2.0 ** -100)])     # (remainders below 2^-126 flush: |x| < ~1e-33 keeps fewer bits)
    w[2, :3] = torch.tensor([16777215.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24])                         # all 24 bits set
    wd = w.to(DEV)
    out = torch.empty(rows * K * 3, dtype=bf16, device=DEV)
    _lib.check(_lib.load().cr_weight_split3(_lib.ctx_for(wd.device), _lib.ptr(wd), _lib.ptr(out), rows, K), "cr_weight_split3")
    torch.cuda.synchronize()
    pl = out.view(rows, K // 32, 3, 4, 8).double().cpu()              # [row][kb][plane][chunk c][e]
    total = pl.sum(2)                                                  # h + m + l, exact in f64
    # chunk c, element e -> k = 32 kb + (4c + e if e < 4 else 16 + 4c + e - 4)
    kk = torch.empty(4, 8, dtype=torch.long)
    for c in range(4):
        for e in range(8):
            kk[c, e] = 4 * c + e if e < 4 else 16 + 4 * c + e - 4
    want = w.double().view(rows, K // 32, 32)[:, :, kk]               # (rows, kb, 4, 8)
    assert torch.equal(total, want), "h + m + l must reproduce every float32 value exactly"
    # plane magnitudes: |m| <= 2^-8 |h|-ish, |l| <= 2^-16 (the dropped products are < 2^-23 of the term)
    h, m, l = pl[:, :, 0], pl[:, :, 1], pl[:, :, 2]
    nz = h.abs() > 1e-30
    assert float((m.abs()[nz] / h.abs()[nz]).max()) <= 2.0 ** -7
    assert float((l.abs()[nz] / h.abs()[nz]).max()) <= 2.0 ** -15


RAW = [  # N, H, W, Cin, Cout, k, stride, pad
    (4, 64, 64, 128, 128, 3, 1, 1),      # 3x3, 128-channel tiles
    (4, 128, 128, 64, 64, 3, 1, 1),      # Cout 64: the 64-wide tiles of both kernels
    (4, 32, 32, 256, 256, 3, 1, 1),      # few tiles: split-K over workgroups + k_splitk_epilogue
    (4, 64, 64, 256, 128, 1, 1, 0),      # 1x1
    (3, 75, 76, 128, 192, 1, 1, 0),      # M tail (17100 pixels), Cout = 3 x 64
    (1, 1, 2048, 1024, 1024, 1, 1, 0),   # FC-shaped: (R,K) x (O,K)^T
    (4, 128, 128, 64, 128, 3, 2, 1),     # stride 2: backward-data by parity class stays on the f32 MFMA
    # the filter-row weight-gradient kernel takes a layer only with >= 64 pixel steps per block: batch sizes chosen for that
    (168, 16, 16, 256, 256, 3, 1, 1),    # W = 16: two edge pixels per 32-pixel step
    (2720, 8, 8, 128, 128, 3, 1, 1),     # W = 8: four edge pixels per step and tap
    (42, 32, 64, 128, 256, 3, 1, 1),     # H != W, Cout = 2 tiles, image boundaries inside the pixel range of a block
]


def _run(prec, x, w, dy, k, st, pd):
    prev = ops.set_precision(prec)
    try:
        N, H, W, Ci = x.shape
        Co = w.shape[0]
        wb, wt = ops.prepared_weights(w, True, f32)
        y = ops.conv_fwd_raw(x, wb, Co, k, st, pd)
        dx = ops.conv_bwd_data_raw(dy, wt, x.shape, k, st, pd) if Ci >= 16 else None
        dw = ops.conv_bwd_weight_raw(dy, x, k, st, pd)
        torch.cuda.synchronize()
        return y, dx, dw
    finally:
        ops.set_precision(prev)


@pytest.mark.parametrize("case", RAW)
def test_split_mode_matches_float64_like_f32_mfma(case):
    N, H, W, Ci, Co, k, st, pd = case
    g = torch.Generator().manual_seed(hash(case) % 2**31)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) * (2.0 / (k * k * Ci)) ** 0.5
    Ho, Wo = (H + 2 * pd - k) // st + 1, (W + 2 * pd - k) // st + 1
    dy = torch.randn(N, Co, Ho, Wo, generator=g)
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, st, pd)
    y64.backward(dy.double())
    xd, dyd = nhwc(x).to(DEV), nhwc(dy).to(DEV)
    wd = w.to(DEV).contiguous(memory_format=torch.channels_last)
    ops.bump_weight_epoch()
    ref = dict(y=y64.detach(), dx=x64.grad, dw=w64.grad)
    errs = {}
    for prec in ("fp32", "fp32x3"):
        y, dx, dw = _run(prec, xd, wd, dyd, k, st, pd)
        errs[prec] = dict(y=relerr(nchw(y), ref["y"]), dx=relerr(nchw(dx), ref["dx"]),
                          dw=relerr(dw, ref["dw"]))
    print(case, errs)
    for key in ("y", "dx", "dw"):
        assert errs["fp32x3"][key] <= 2.0 * errs["fp32"][key] + 3e-7, (key, errs)
        assert errs["fp32x3"][key] < 2e-5


def test_layers_outside_the_split_kernels_fall_back_to_f32_mfma():
    """k extent not a multiple of 32 (the stem) and tiny maps: identical bits in both modes"""
    g = torch.Generator().manual_seed(3)
    for (N, H, W, Ci, Co, k, st, pd) in [(2, 32, 32, 4, 16, 7, 1, 3), (2, 32, 32, 16, 32, 3, 2, 1), (1, 8, 8, 64, 64, 3, 1, 1)]:
        x = nhwc(torch.randn(N, Ci, H, W, generator=g)).to(DEV)
        w = (torch.randn(Co, Ci, k, k, generator=g) * 0.1).to(DEV).contiguous(memory_format=torch.channels_last)
        Ho, Wo = (H + 2 * pd - k) // st + 1, (W + 2 * pd - k) // st + 1
        dy = nhwc(torch.randn(N, Co, Ho, Wo, generator=g)).to(DEV)
        a = _run("fp32", x, w, dy, k, st, pd)
        b = _run("fp32x3", x, w, dy, k, st, pd)
        assert torch.equal(a[0], b[0])
        if Ci >= 16:
            assert torch.equal(a[1], b[1])


def test_row_kernel_bias_gradient_and_accumulate():
    """the filter-row weight-gradient kernel (3x3, 128-multiple channels): bias gradient from the staged dy planes, and
    accumulation into an existing gradient buffer (the sink of the flat gradient)"""
    g = torch.Generator().manual_seed(9)
    N, H, W, C = 176, 32, 32, 128                      # 3 row tiles -> 85 splits x >= 64 steps
    x = torch.randn(N, C, H, W, generator=g); dy = torch.randn(N, C, H, W, generator=g)
    w64 = torch.zeros(C, C, 3, 3, dtype=f64, requires_grad=True)
    y = F.conv2d(x.double(), w64, None, 1, 1)
    y.backward(dy.double())
    prev = ops.set_precision("fp32x3")
    try:
        xd, dyd = nhwc(x).to(DEV), nhwc(dy).to(DEV)
        sink = torch.ones(C, C, 3, 3, device=DEV).contiguous(memory_format=torch.channels_last)
        bias = torch.full((C,), 2.0, device=DEV)
        ops.conv_bwd_weight_raw(dyd, xd, 3, 1, 1, sink=sink, bias_acc=bias)
        torch.cuda.synchronize()
    finally:
        ops.set_precision(prev)
    assert relerr(sink - 1.0, w64.grad) < 2e-6
    assert relerr(bias - 2.0, dy.double().sum((0, 2, 3))) < 2e-6
