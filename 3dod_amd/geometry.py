"""Python entry points of the geometry kernels (thin: argument checking, output
allocation, one C-ABI call each).  Semantics documented in include/cr3dod.h."""
import os

import torch

from . import _lib

f32 = torch.float32


def _f32c(t, name, shape=None):
    if t.dtype != f32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if not t.is_cuda:
        raise _lib.CrError(f"{name} must be a CUDA(HIP) tensor; 3dod_amd has no CPU path")
    if shape is not None:
        if t.dim() != len(shape) or any(s is not None and s != d for s, d in zip(shape, t.shape)):
            raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {shape}")
    return t.contiguous()


def cuboid_corners(box6, R):
    """get_cuboid_verts_faces (verts) -- cubercnn/util/math_util.py:142-245."""
    box6 = _f32c(box6, "box6", (None, 6))
    n = box6.shape[0]
    R = _f32c(R, "R", (n, 3, 3))
    verts = torch.empty((n, 8, 3), dtype=f32, device=box6.device)
    lib = _lib.load()
    _lib.check(lib.cr_cuboid_corners(_lib.ctx_for(box6.device), _lib.ptr(box6), _lib.ptr(R), n, _lib.ptr(verts)),
               "cr_cuboid_corners")
    return verts


def cubes_project_score(cubes, K, im_wh, ref_boxes, prior_mu, prior_sigma, rect_pts=None,
                        want=("corners", "boxes", "iou", "dim", "corner", "combined"), iou_boxes=None, fast=None, stats=None,
                        out=None):
    """Fused K17 (see cr_cubes_project_score).  Returns a dict with the requested
    planes plus `argmax` (N,) int64 and `best` (N,).  iou_boxes (N,4): the box of the IoU term when it is not
    ref_boxes (the GT-box branches of ROIHeads_Boxer score IoU against the projected ground-truth cube).
    fast: cr_cubes_project_score_fast (argmax / best bit-equal, planes to 1e-4).  None = fast exactly when no plane is
    requested -- then nothing that leaves the kernel differs (CR_GEO_EXACT=1: always the exact kernel).  stats: int64 (2,)
    device tensor, fast kernel only: [0] += objects that took the exact sequence, [1] += re-evaluated candidates.
    out: the dict a previous call with the same shapes and `want` returned -- its tensors are written again instead of
    allocating eight new ones."""
    if fast is None:
        fast = len(want) == 0 and os.environ.get("CR_GEO_EXACT", "0") != "1"
    cubes = _f32c(cubes, "cubes", (None, None, 15))
    N, Pn = cubes.shape[:2]
    dev = cubes.device
    K = _f32c(K, "K")
    if K.shape == (3, 3):
        kpo = 0
    elif K.shape == (N, 3, 3):
        kpo = 1
    else:
        raise ValueError(f"K must be (3,3) or ({N},3,3), got {tuple(K.shape)}")
    ref_boxes = _f32c(ref_boxes, "ref_boxes", (N, 4))
    prior_mu = _f32c(prior_mu, "prior_mu", (N, 3))
    prior_sigma = _f32c(prior_sigma, "prior_sigma", (N, 3))
    if rect_pts is not None:
        rect_pts = _f32c(rect_pts, "rect_pts", (N, 4, 2))
    if iou_boxes is not None:
        iou_boxes = _f32c(iou_boxes, "iou_boxes", (N, 4))
    shapes = {"corners": (N, Pn, 8, 2), "boxes": (N, Pn, 4), "iou": (N, Pn), "dim": (N, Pn),
              "corner": (N, Pn), "combined": (N, Pn)}
    if out is not None:
        ok = all((out.get(k) is None) == (k not in want) and (out.get(k) is None or (tuple(out[k].shape) == shapes[k] and
                 out[k].dtype == f32 and out[k].device == dev and out[k].is_contiguous())) for k in shapes)
        if not ok or tuple(out["argmax"].shape) != (N,) or tuple(out["best"].shape) != (N,) or out["argmax"].device != dev:
            raise ValueError("out: not the result of a call with these shapes and this `want`")
    else:
        out = {k: (torch.empty(shapes[k], dtype=f32, device=dev) if k in want else None) for k in shapes}
        out["argmax"] = torch.empty((N,), dtype=torch.int64, device=dev)        # the kernel writes every object's entry
        out["best"] = torch.empty((N,), dtype=f32, device=dev)
    if N == 0:
        return out
    lib = _lib.load()
    extra = ()
    if fast:
        if stats is not None and (stats.dtype != torch.int64 or stats.numel() != 2 or not stats.is_cuda):
            raise ValueError("stats must be an int64 device tensor of 2 elements")
        extra = (_lib.ptr(stats),)
    fn = lib.cr_cubes_project_score_fast if fast else lib.cr_cubes_project_score
    rc = fn(
        _lib.ctx_for(dev), _lib.ptr(cubes), N, Pn, _lib.ptr(K), kpo, float(im_wh[0]), float(im_wh[1]),
        _lib.ptr(ref_boxes), _lib.ptr(prior_mu), _lib.ptr(prior_sigma), _lib.ptr(rect_pts),
        _lib.ptr(out["corners"]), _lib.ptr(out["boxes"]), _lib.ptr(out["iou"]), _lib.ptr(out["dim"]),
        _lib.ptr(out["corner"]), _lib.ptr(out["combined"]), _lib.ptr(out["argmax"]), _lib.ptr(out["best"]),
        _lib.ptr(iou_boxes), *extra)
    _lib.check(rc, "cr_cubes_project_score")
    return out


def propose_from_draws(boxes, depth, prior_mu, prior_sigma, K, P, dim_normals, ctr_normals, yaw_idx, normal):
    """K18 with caller-supplied variates (see cr_propose).  Returns (cubes (N,P,15), exhausted int tensor)."""
    boxes = _f32c(boxes, "boxes", (None, 4))
    N = boxes.shape[0]
    dev = boxes.device
    depth = _f32c(depth, "depth", (None, None))
    H, W = depth.shape
    prior_mu = _f32c(prior_mu, "prior_mu", (N, 3))
    prior_sigma = _f32c(prior_sigma, "prior_sigma", (N, 3))
    K = _f32c(K, "K", (3, 3))
    dim_normals = _f32c(dim_normals, "dim_normals", (None, 3, N, P))
    ctr_normals = _f32c(ctr_normals, "ctr_normals", (3, N, P))
    if yaw_idx.dtype != torch.int32:
        yaw_idx = yaw_idx.to(torch.int32)
    yaw_idx = yaw_idx.contiguous()
    assert tuple(yaw_idx.shape) == (N, P)
    normal = _f32c(normal, "normal", (3,))
    cubes = torch.empty((N, P, 15), dtype=f32, device=dev)
    exhausted = torch.zeros((1,), dtype=torch.int32, device=dev)
    lib = _lib.load()
    rc = lib.cr_propose(_lib.ctx_for(dev), _lib.ptr(boxes), N, _lib.ptr(depth), H, W, _lib.ptr(prior_mu),
                        _lib.ptr(prior_sigma), _lib.ptr(K), P, _lib.ptr(dim_normals), dim_normals.shape[0],
                        _lib.ptr(ctr_normals), _lib.ptr(yaw_idx), _lib.ptr(normal), _lib.ptr(cubes),
                        _lib.ptr(exhausted))
    _lib.check(rc, "cr_propose")
    return cubes, exhausted


def propose_from_draws_batched(boxes, img_idx, depth, prior_mu, prior_sigma, K, P, dim_normals, ctr_normals, yaw_idx, normals):
    """cr_propose_batched: the objects of B images in one launch.  boxes (N,4), img_idx (N) int32, depth (B,H,W),
    K (B,3,3), normals (B,3); draws as in propose_from_draws.  Returns (cubes (N,P,15), exhausted int tensor)."""
    boxes = _f32c(boxes, "boxes", (None, 4))
    N, dev = boxes.shape[0], boxes.device
    depth = _f32c(depth, "depth", (None, None, None))
    B, H, W = depth.shape
    K = _f32c(K, "K", (B, 3, 3))
    normals = _f32c(normals, "normals", (B, 3))
    prior_mu = _f32c(prior_mu, "prior_mu", (N, 3))
    prior_sigma = _f32c(prior_sigma, "prior_sigma", (N, 3))
    dim_normals = _f32c(dim_normals, "dim_normals", (None, 3, N, P))
    ctr_normals = _f32c(ctr_normals, "ctr_normals", (3, N, P))
    yaw_idx = yaw_idx.to(torch.int32).contiguous()
    img_idx = img_idx.to(device=dev, dtype=torch.int32).contiguous()
    assert tuple(yaw_idx.shape) == (N, P) and tuple(img_idx.shape) == (N,)
    cubes = torch.empty((N, P, 15), dtype=f32, device=dev)
    exhausted = torch.zeros((1,), dtype=torch.int32, device=dev)
    lib = _lib.load()
    _lib.check(lib.cr_propose_batched(_lib.ctx_for(dev), _lib.ptr(boxes), _lib.ptr(img_idx), N, _lib.ptr(depth), B, H, W,
                                      _lib.ptr(prior_mu), _lib.ptr(prior_sigma), _lib.ptr(K), P, _lib.ptr(dim_normals),
                                      dim_normals.shape[0], _lib.ptr(ctr_normals), _lib.ptr(yaw_idx), _lib.ptr(normals),
                                      _lib.ptr(cubes), _lib.ptr(exhausted)), "cr_propose_batched")
    return cubes, exhausted


def ransac_plane_batched(pts, triples, eligible=None, thresh=0.05):
    """B plane fits at once (cr_ransac_plane_batched): pts (B,Q,3), triples (B,T,3) int32 indices of eligible points,
    eligible (B,Q) bool/uint8 or None.  Returns (-equation (B,4), counts (B,T), best (B,2) = idx,count).  The caller
    guarantees the triples are in range (they come from the device-side sampler; checking them would be a sync)."""
    pts = _f32c(pts, "pts", (None, None, 3))
    B, Q = pts.shape[0], pts.shape[1]
    triples = triples.to(torch.int32).contiguous()
    T = triples.shape[1]
    assert tuple(triples.shape) == (B, T, 3) and triples.is_cuda
    dev = pts.device
    if eligible is not None:
        eligible = eligible.to(torch.uint8).contiguous()
        assert tuple(eligible.shape) == (B, Q)
    neg_eq = torch.empty((B, 4), dtype=f32, device=dev)
    counts = torch.empty((B, T), dtype=torch.int32, device=dev)
    best = torch.empty((B, 2), dtype=torch.int32, device=dev)
    lib = _lib.load()
    _lib.check(lib.cr_ransac_plane_batched(_lib.ctx_for(dev), _lib.ptr(pts), _lib.ptr(eligible), B, Q, _lib.ptr(triples), T,
                                           float(thresh), _lib.ptr(neg_eq), _lib.ptr(counts), _lib.ptr(best)),
               "cr_ransac_plane_batched")
    return neg_eq, counts, best


def ransac_plane(pts, triples, thresh=0.05, validate=True):
    """K21 Plane.fit_parallel with given triples.  Returns (-equation (4,), counts (T,), best (2,) = idx,count)."""
    pts = _f32c(pts, "pts", (None, 3))
    Q = pts.shape[0]
    if triples.dtype != torch.int32:
        triples = triples.to(torch.int32)
    triples = triples.contiguous()
    T = triples.shape[0]
    assert triples.shape == (T, 3) and triples.is_cuda
    if validate:      # an out-of-range index would be an out-of-bounds device read
        mn, mx = int(triples.min()), int(triples.max())
        if mn < 0 or mx >= Q:
            raise ValueError(f"triples index outside [0,{Q})")
    dev = pts.device
    neg_eq = torch.empty((4,), dtype=f32, device=dev)
    counts = torch.empty((T,), dtype=torch.int32, device=dev)
    best = torch.empty((2,), dtype=torch.int32, device=dev)
    lib = _lib.load()
    rc = lib.cr_ransac_plane(_lib.ctx_for(dev), _lib.ptr(pts), Q, _lib.ptr(triples), T, float(thresh),
                             _lib.ptr(neg_eq), _lib.ptr(counts), _lib.ptr(best))
    _lib.check(rc, "cr_ransac_plane")
    return neg_eq, counts, best


def box3d_overlap(boxes1, boxes2):
    """exact intersection volume and IoU of oriented 3D boxes: boxes1 (N,8,3), boxes2 (M,8,3) corners (pytorch3d order,
    as produced by get_cuboid_verts_faces) -> (vol (N,M), iou (N,M)).  Stands in for pytorch3d.ops.box3d_overlap at
    ProposalNetwork/utils/utils.py:207 and cubercnn/evaluation/omni3d_evaluation.py:155."""
    lib = _lib.load()
    if not boxes1.is_cuda:
        raise _lib.CrError("box3d_overlap: expected CUDA(HIP) tensors; 3dod_amd has no CPU path")
    b1, b2 = boxes1.float().contiguous(), boxes2.float().contiguous()
    N, M = b1.shape[0], b2.shape[0]
    vol = torch.empty((N, M), dtype=torch.float32, device=b1.device)
    iou = torch.empty((N, M), dtype=torch.float32, device=b1.device)
    _lib.check(lib.cr_box3d_overlap(_lib.ctx_for(b1.device), _lib.ptr(b1), _lib.ptr(b2), N, M, _lib.ptr(vol), _lib.ptr(iou)),
               "cr_box3d_overlap")
    return vol, iou


def box_median(depth, boxes, img):
    """lower median (torch.median) of depth[img[i], y1:y2, x1:x2]: depth (B,H,W) f32, boxes (n,4) int32 (x1,y1,x2,y2),
    img (n) int32 -> (n,) f32, NaN for an empty window.  One radix-select block per box (cr_box_median) instead of the
    per-box loop of cubercnn/modeling/roi_heads/roi_heads.py:1216-1218."""
    lib = _lib.load()
    if not depth.is_cuda:
        raise _lib.CrError("box_median: expected CUDA(HIP) tensors; 3dod_amd has no CPU path")
    d = depth.float().contiguous()
    assert d.dim() == 3 and boxes.dim() == 2 and boxes.shape[1] == 4 and img.shape[0] == boxes.shape[0]
    b, im = boxes.to(torch.int32).contiguous(), img.to(torch.int32).contiguous()
    out = torch.empty(b.shape[0], dtype=torch.float32, device=d.device)
    _lib.check(lib.cr_box_median(_lib.ctx_for(d.device), _lib.ptr(d), d.shape[0], d.shape[1], d.shape[2], _lib.ptr(b),
                                 _lib.ptr(im), b.shape[0], _lib.ptr(out)), "cr_box_median")
    return out


def hull8(points):
    """convex hull of each RoI's 8 projected corners in the reference's order (cr_hull8): points (n,8,2) f32 ->
    order (n,8) int64, count (n) int32, bump (n,8) f32"""
    lib = _lib.load()
    if not points.is_cuda:
        raise _lib.CrError("hull8: expected CUDA(HIP) tensors; 3dod_amd has no CPU path")
    p = points.detach().float().contiguous()
    n = p.shape[0]
    order = torch.empty((n, 8), dtype=torch.int32, device=p.device)
    count = torch.empty((n,), dtype=torch.int32, device=p.device)
    bump = torch.empty((n, 8), dtype=torch.float32, device=p.device)
    _lib.check(lib.cr_hull8(_lib.ctx_for(p.device), _lib.ptr(p), n, _lib.ptr(order), _lib.ptr(count), _lib.ptr(bump)), "cr_hull8")
    return order.long(), count, bump


class _PolygonFocal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hull, count, masks, mask_idx):
        lib = _lib.load()
        h = hull.detach().float().contiguous()
        n = h.shape[0]
        loss = torch.empty((n,), dtype=torch.float32, device=h.device)
        grad = torch.empty((n, 8, 2), dtype=torch.float32, device=h.device) if hull.requires_grad else None
        ones = (masks != 0).sum((1, 2)).to(torch.int32).contiguous()
        _lib.check(lib.cr_polygon_focal(_lib.ctx_for(h.device), _lib.ptr(h), _lib.ptr(count), _lib.ptr(masks), _lib.ptr(mask_idx),
                                        _lib.ptr(ones), n, masks.shape[1], masks.shape[2], _lib.ptr(loss),
                                        _lib.ptr(grad) if grad is not None else None), "cr_polygon_focal")
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, g):
        return (ctx.grad * g.view(-1, 1, 1)) if ctx.grad is not None else None, None, None, None


def polygon_focal(hull, count, masks, mask_idx):
    """segment_loss per RoI (cr_polygon_focal): hull (n,8,2) ordered vertices (differentiable), count (n) int32,
    masks (Nm,H,W) uint8, mask_idx (n) int32 -> (n,) float32"""
    if not hull.is_cuda:
        raise _lib.CrError("polygon_focal: expected CUDA(HIP) tensors; 3dod_amd has no CPU path")
    assert masks.dtype == torch.uint8 and masks.dim() == 3 and masks.is_contiguous()
    return _PolygonFocal.apply(hull, count.to(torch.int32).contiguous(), masks, mask_idx.to(torch.int32).contiguous())


def segment_counts(corners2d, mask, stride=4):
    """per proposal: samples (stride*i, stride*j) inside the filled hull of its 8 projected corners, and how many of
    those the object mask covers (cr_segment_counts): corners2d (P,8,2) f32, mask (H,W) bool/uint8 -> (P,2) int64"""
    lib = _lib.load()
    if not corners2d.is_cuda:
        raise _lib.CrError("segment_counts: expected CUDA(HIP) tensors; 3dod_amd has no CPU path")
    c = corners2d.detach().float().contiguous()
    m = mask.to(device=c.device, dtype=torch.uint8).contiguous()
    P = c.shape[0]
    out = torch.empty((P, 2), dtype=torch.int32, device=c.device)
    _lib.check(lib.cr_segment_counts(_lib.ctx_for(c.device), _lib.ptr(c), P, _lib.ptr(m), m.shape[0], m.shape[1], int(stride),
                                     _lib.ptr(out)), "cr_segment_counts")
    return out.long()


_scratch = {}


def _workspace(dev, nbytes):
    """grow-only per-device scratch for kernels that need O(input) temporary storage: a fresh 100+ MB request per call
    makes the caching allocator split and re-grow its large blocks (a ~20 ms hipMalloc every few calls).  Used on the
    current stream only."""
    buf = _scratch.get(dev)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _scratch.pop(dev, None)
        buf = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        _scratch[dev] = buf
    return buf


def mask_rects(masks):
    """minimum-area rectangle of the largest 8-connected component of every mask (cr_mask_rects; the cv2 step of
    score_corners, scorefunction.py:58-68): masks (n,H,W) bool/uint8 on the GPU, or a list of such tensors (the masks of
    several images: passed as a pointer table, not concatenated) -> rects (n,4,2) f32 (a NaN row for an empty mask),
    valid (n) bool"""
    lib = _lib.load()
    many = isinstance(masks, (list, tuple))
    ms = [m for m in (masks if many else [masks])]
    if not all(m.is_cuda for m in ms):
        raise _lib.CrError("mask_rects: expected CUDA(HIP) tensors; 3dod_amd has no CPU path")
    ms = [m.contiguous() for m in ms]
    ms = [m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8) for m in ms]
    H, W = ms[0].shape[-2:]
    assert all(m.dim() == 3 and tuple(m.shape[-2:]) == (H, W) for m in ms)
    n = sum(m.shape[0] for m in ms)
    dev = ms[0].device
    px = n * H * W
    ws = _workspace(dev, 8 * px + 32 * n + 64)
    labels, sizes = ws[:4 * px], ws[4 * px:8 * px]
    tail = ws[8 * px + (-8 * px) % 16:]
    best, bbox = tail[:8 * n], tail[8 * n:24 * n]
    rects = torch.empty((n, 4, 2), dtype=torch.float32, device=dev)
    valid = torch.empty((n,), dtype=torch.uint8, device=dev)
    dense, table = (ms[0], None) if len(ms) == 1 else (None, None)
    if dense is None:
        import numpy as np
        table = torch.from_numpy(np.concatenate([m.data_ptr() + np.arange(m.shape[0], dtype=np.int64) * (H * W) for m in ms]))
        table = table.to(dev)
    _lib.check(lib.cr_mask_rects(_lib.ctx_for(dev), _lib.ptr(dense), _lib.ptr(table), n, H, W, _lib.ptr(labels), _lib.ptr(sizes),
                                 _lib.ptr(best), _lib.ptr(bbox), _lib.ptr(rects), _lib.ptr(valid)), "cr_mask_rects")
    return rects, valid.bool()
