"""Kernels of the Depth-Anything-V2 forward against plain PyTorch float32 references of the same ops (the inputs are the
bf16 values the kernel sees; tolerances are those of bf16 storage: the attention probabilities and every output are
rounded to bf16)."""
import importlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ops = importlib.import_module("3dod_amd.hipops")
DEV = torch.device("cuda:0")


def relerr(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


@pytest.mark.parametrize("B,N,H", [(1, 1, 1), (2, 17, 2), (1, 64, 3), (2, 65, 2), (1, 197, 6), (2, 1370, 16), (1, 2000, 4)])
def test_attention(B, N, H):
    D = 64
    g = torch.Generator().manual_seed(N * 7 + H)
    qkv = (torch.randn(B * N, 3 * H * D, generator=g) * 1.5).to(torch.bfloat16)
    out = ops.attention(qkv.to(DEV), B, N, H, D, D ** -0.5).cpu()
    q, k, v = qkv.float().view(B, N, 3, H, D).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax((q * D ** -0.5) @ k.transpose(-2, -1), dim=-1) @ v).transpose(1, 2).reshape(B * N, H * D)
    assert out.shape == ref.shape and torch.isfinite(out.float()).all()
    assert relerr(out, ref) < 1.5e-2, relerr(out, ref)
    assert float((out.float() - ref).abs().max()) < 0.05 * float(ref.abs().max())


def test_attention_peaked_and_shifted_scores():
    """large logits (one dominant key) and a constant offset: the online softmax must neither overflow nor lose the row"""
    B, N, H, D = 1, 300, 2, 64
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(B * N, 3 * H * D, generator=g)
    qkv.view(B, N, 3, H, D)[:, :, 0] *= 12.0
    qkv = qkv.to(torch.bfloat16)
    out = ops.attention(qkv.to(DEV), B, N, H, D, D ** -0.5).cpu()
    q, k, v = qkv.float().view(B, N, 3, H, D).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax((q * D ** -0.5) @ k.transpose(-2, -1), dim=-1) @ v).transpose(1, 2).reshape(B * N, H * D)
    assert torch.isfinite(out.float()).all() and relerr(out, ref) < 2e-2


def test_attention_rejects_other_head_dims():
    lib = importlib.import_module("3dod_amd._lib")
    with pytest.raises(lib.CrError):
        ops.attention(torch.zeros(4, 3 * 2 * 32, dtype=torch.bfloat16, device=DEV), 1, 4, 2, 32, 1.0)


@pytest.mark.parametrize("M,C", [(1, 64), (37, 384), (1370, 1024), (5, 1536)])
def test_layernorm(M, C):
    g = torch.Generator().manual_seed(M + C)
    x = (torch.randn(M, C, generator=g) * 3 + 1.5).to(torch.bfloat16)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    y = ops.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-6).cpu()
    ref = F.layer_norm(x.float(), (C,), gamma, beta, 1e-6)
    assert relerr(y, ref) < 4e-3
    assert float((y.float() - ref).abs().max()) < 0.03 * float(ref.abs().max())


def test_gelu_and_scale_residual():
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(333, 128, generator=g) * 3).to(torch.bfloat16)
    y = ops.gelu_(x.to(DEV).clone()).cpu()
    assert relerr(y, F.gelu(x.float())) < 3e-3
    a = torch.randn(77, 256, generator=g).to(torch.bfloat16)
    b = torch.randn(77, 256, generator=g).to(torch.bfloat16)
    gam = torch.randn(256, generator=g)
    assert relerr(ops.scale_residual(a.to(DEV), b.to(DEV), gam.to(DEV)).cpu(), a.float() + gam * b.float()) < 3e-3
    assert relerr(ops.scale_residual(a.to(DEV), b.to(DEV)).cpu(), a.float() + b.float()) < 3e-3


@pytest.mark.parametrize("h,w,Ho,Wo,C", [(7, 10, 14, 20, 32), (37, 37, 74, 74, 64), (5, 9, 13, 4, 16), (1, 1, 3, 3, 8),
                                          (148, 148, 518, 518, 16)])
def test_resize_bilinear_align_corners(h, w, Ho, Wo, C):
    g = torch.Generator().manual_seed(h * w)
    x = torch.randn(2, h, w, C, generator=g).to(torch.bfloat16)
    y = ops.resize_bilinear_ac(x.to(DEV), (Ho, Wo)).cpu()
    ref = F.interpolate(x.float().permute(0, 3, 1, 2), (Ho, Wo), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    assert y.shape == ref.shape
    assert float((y.float() - ref).abs().max()) < 0.02 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("M,C", [(3, 64), (1370, 1024), (9, 1536)])
def test_fused_residual_layernorm_equals_the_two_kernels(M, C):
    g = torch.Generator().manual_seed(C)
    x = torch.randn(M, C, generator=g).to(torch.bfloat16).to(DEV)
    y = torch.randn(M, C, generator=g).to(torch.bfloat16).to(DEV)
    ls, gam, bet = (torch.randn(C, generator=g).to(DEV) for _ in range(3))
    xo, ho = ops.scale_residual_layernorm(x, y, ls, gam, bet, 1e-6)
    x2 = ops.scale_residual(x, y, ls)
    h2 = ops.layernorm(x2, gam, bet, 1e-6)
    assert torch.equal(xo, x2) and torch.equal(ho, h2)
    xo, ho = ops.scale_residual_layernorm(x, y, None, gam, bet, 1e-6)
    assert torch.equal(xo, ops.scale_residual(x, y)) and torch.equal(ho, ops.layernorm(xo, gam, bet, 1e-6))
