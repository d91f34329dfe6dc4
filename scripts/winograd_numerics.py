"""Winograd F(2x2, 3x3) in float32 against direct float32 convolution, both measured against float64: what the 2.25x
reduction of multiplies on the 3x3 256 -> 256 layers would cost in accuracy (CPU, torch).  Also F(4x4, 3x3) (4x)."""
import torch
torch.manual_seed(0)
N, C, K, H, W = 2, 256, 256, 32, 32
x = torch.randn(N, C, H, W) * 0.7
w = torch.randn(K, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
direct = torch.nn.functional.conv2d(x, w, padding=1)


def winograd(x, w, m):
    if m == 2:
        BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1.]])
        G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1.]])
        AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1.]])
    else:
        BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                           [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1.]])
        G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                          [1 / 24, -1 / 12, 1 / 6], [0, 0, 1.]])
        AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1.]])
    a = m + 2
    xp = torch.nn.functional.pad(x, (1, 1 + (-x.shape[3]) % m, 1, 1 + (-x.shape[2]) % m))
    t = xp.unfold(2, a, m).unfold(3, a, m)                                   # N, C, th, tw, a, a
    V = torch.einsum("ij,nchwjk,lk->nchwil", BT, t, BT)
    U = torch.einsum("ij,kcjl,ml->kcim", G, w, G)
    M = torch.einsum("nchwil,kcil->nkhwil", V, U)
    Y = torch.einsum("ij,nkhwjl,ml->nkhwim", AT, M, AT)                      # N, K, th, tw, m, m
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(x.shape[0], w.shape[0], Y.shape[2] * m, Y.shape[3] * m)
    return y[:, :, :x.shape[2], :x.shape[3]]


scale = ref.abs().max()
for name, y in (("direct f32", direct), ("winograd F(2x2,3x3) f32", winograd(x, w, 2)), ("winograd F(4x4,3x3) f32", winograd(x, w, 4))):
    e = (y.double() - ref).abs()
    print(f"{name:26s} max err / max|y| = {float(e.max() / scale):.2e}   rms err / rms y = {float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e}")
