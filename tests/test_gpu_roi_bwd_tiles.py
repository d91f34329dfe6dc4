"""cr_roi_align_bwd_set (csrc/detection.hip: k_roi_bbox + k_roi_bwd_tiles) -- the RoIAlign backward without global atomics:
one block owns a 16 x 16-pixel tile x 64 channels of a gradient map, sums the RoIs that reach it in RoI order and writes
every pixel once.  Checked against the vectorised CPU restatement of torchvision's roi_align + detectron2's level assignment
(oracle/cpu_backend.roi_align_pyramid through autograd; roi_heads.py:2178,2273 of the reference are the call sites), against
the atomic kernel it replaces, for bit-reproducibility, and on the edge cases the forward defines (boxes outside the image,
NaN / empty boxes, RoIs larger than the map, zero RoIs: the maps must still be written -- with zeros)."""
import ctypes
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu
ops = importlib.import_module("3dod_amd.hipops")
lib_mod = importlib.import_module("3dod_amd._lib")
DEV = torch.device("cuda:0")
f32 = torch.float32
SCALES = [1 / 4, 1 / 8, 1 / 16, 1 / 32, 1 / 64]


def _rois(n_img, R, size, g, clustered=True):
    """Omni3D-shaped RoIs: clusters of near-duplicate proposals around a few centres per image, all pyramid levels"""
    img = torch.randint(0, n_img, (R,), generator=g).float()
    k = 12
    centres = torch.rand(n_img, k, 2, generator=g) * size
    which = torch.randint(0, k, (R,), generator=g)
    ctr = centres[img.long(), which] + torch.randn(R, 2, generator=g) * (8 if clustered else size / 4)
    wh = torch.exp(torch.rand(R, 2, generator=g) * 3.2 + 2.5)                  # 12 ... 300 px
    b = torch.cat([ctr - wh / 2, ctr + wh / 2], 1).clamp(0, size - 1)
    return torch.cat([img[:, None], b], 1)


def _run(kind, shapes, rois, dout, C, grads=None):
    """kind: 'tiles' | 'atomic'; returns the list of gradient maps (NHWC f32)"""
    lib = lib_mod.load()
    n = len(shapes)
    if grads is None:
        grads = [(torch.full if kind == "tiles" else torch.zeros)(s, *((7.0,) if kind == "tiles" else ()), dtype=f32, device=DEV)
                 for s in shapes]                                           # tiles: garbage in, every pixel must be overwritten
    ptrs = (ctypes.c_void_p * n)(*[g.data_ptr() for g in grads])
    Hs = (ctypes.c_int * n)(*[s[1] for s in shapes])
    Ws = (ctypes.c_int * n)(*[s[2] for s in shapes])
    sc = (ctypes.c_float * n)(*SCALES[:n])
    cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
    ctx = lib_mod.ctx_for(DEV)
    af = 1 if dout.dtype == f32 else 0
    if kind == "tiles":
        lib_mod.check(lib.cr_roi_align_bwd_set(ctx, cast(ptrs), cast(Hs), cast(Ws), cast(sc), n, C, shapes[0][0], lib_mod.ptr(rois),
                                               rois.shape[0], 7, 7, lib_mod.ptr(dout), af), "cr_roi_align_bwd_set")
    else:
        lib_mod.check(lib.cr_roi_align_bwd(ctx, cast(ptrs), cast(Hs), cast(Ws), cast(sc), n, C, lib_mod.ptr(rois), rois.shape[0],
                                           7, 7, lib_mod.ptr(dout), af), "cr_roi_align_bwd")
    torch.cuda.synchronize()
    return grads


def _oracle(shapes, rois, dout):
    from oracle import cpu_backend as O
    feats = [torch.zeros(s, dtype=f32, requires_grad=True) for s in shapes]
    y = O.roi_align_pyramid(feats, rois, SCALES[:len(shapes)], 7)
    y.backward(dout)
    return [f.grad if f.grad is not None else torch.zeros(s) for f, s in zip(feats, shapes)]


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-20))


@pytest.mark.parametrize("N,size,R,C", [(2, 256, 300, 64), (4, 512, 2048, 256)])
def test_tiles_match_oracle_and_atomic_kernel_and_are_reproducible(N, size, R, C):
    g = torch.Generator().manual_seed(R)
    shapes = [(N, size // s, size // s, C) for s in (4, 8, 16, 32, 64)]
    rois = _rois(N, R, size, g)
    dout = torch.randn(R, 7, 7, C, generator=g)
    dout[::7] = 0                                                           # masked (padding) RoI slots carry zero gradient
    a = _run("tiles", shapes, rois.to(DEV), dout.to(DEV), C)
    b = _run("atomic", shapes, rois.to(DEV), dout.to(DEV), C)
    a2 = _run("tiles", shapes, rois.to(DEV), dout.to(DEV), C)
    for x, y in zip(a, a2):
        assert torch.equal(x, y), "the tile-owner backward must be bit-reproducible"
    for x, y in zip(a, b):
        assert _rel(x, y) < 2e-6, _rel(x, y)                                # same sums, other association order
    if R <= 300:
        ref = _oracle(shapes, rois, dout)
        for x, y in zip(a, ref):
            assert _rel(x.cpu(), y) < 1e-5, _rel(x.cpu(), y)


def test_tiles_edge_cases():
    C, N, size = 64, 2, 192                                                  # 48 x 48 ... 3 x 3 maps: partial tiles on every level
    shapes = [(N, size // s, size // s, C) for s in (4, 8, 16, 32, 64)]
    nan = float("nan")
    rois = torch.tensor([[0, 10., 10, 60, 70], [1, -300, -300, -200, -250],      # far outside: no sample
                         [0, -20, -30, 40, 50], [1, 150, 160, 400, 420],         # crossing the border
                         [0, 50, 50, 50, 50], [1, 30, 40, 30.5, 41],            # empty / sub-pixel
                         [0, nan, 0, 50, 50], [1, 0, 0, 191, 191],              # NaN box; whole image (top level, one tile)
                         [0, 0, 0, 191, 191], [1, 96, 0, 97, 191]])             # very elongated
    g = torch.Generator().manual_seed(3)
    dout = torch.randn(rois.shape[0], 7, 7, C, generator=g)
    a = _run("tiles", shapes, rois.to(DEV), dout.to(DEV), C)
    b = _run("atomic", shapes, rois.to(DEV), dout.to(DEV), C)
    finite = torch.isfinite(rois).all(1)
    ref = _oracle(shapes, rois[finite], dout[finite])
    for x, y, z in zip(a, b, ref):
        assert torch.isfinite(x).all()
        assert _rel(x, y) < 2e-6 and _rel(x.cpu(), z) < 1e-5
    # no RoI at all: every map is written with zeros (the caller does not zero-fill for this kernel)
    z = _run("tiles", shapes, torch.zeros(0, 5, device=DEV), torch.zeros(0, 7, 7, C, device=DEV), C)
    assert all(float(m.abs().max()) == 0.0 for m in z)


def test_tiles_bf16_gradient_and_autograd_route():
    """bf16 mode: dY arrives as bf16; and the autograd wrapper takes this kernel (no zero fill) when the shape allows"""
    g = torch.Generator().manual_seed(9)
    N, size, R, C = 2, 256, 200, 128
    shapes = [(N, size // s, size // s, C) for s in (4, 8, 16, 32, 64)]
    rois = _rois(N, R, size, g)
    dout = torch.randn(R, 7, 7, C, generator=g).to(torch.bfloat16)
    a = _run("tiles", shapes, rois.to(DEV), dout.to(DEV), C)
    ref = _oracle(shapes, rois, dout.float())
    for x, y in zip(a, ref):
        assert _rel(x.cpu(), y) < 1e-5
    feats = [torch.randn(s, generator=g).to(DEV).requires_grad_(True) for s in shapes]
    y = ops.roi_align_pyramid(feats, rois.to(DEV), SCALES, 7)
    y.backward(dout.float().to(DEV))
    prev = ops._ROI_BWD_TILES[0]
    try:
        ops._ROI_BWD_TILES[0] = False
        feats2 = [f.detach().clone().requires_grad_(True) for f in feats]
        ops.roi_align_pyramid(feats2, rois.to(DEV), SCALES, 7).backward(dout.float().to(DEV))
    finally:
        ops._ROI_BWD_TILES[0] = prev
    for f1, f2 in zip(feats, feats2):
        assert _rel(f1.grad, f2.grad) < 2e-6
