"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the
1000-cube proposal-and-scoring geometry of luchsonice/3dod in numpy float32.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (3dod_amd/) never does.

Every function cites the reference file:line it restates (paths are into the
reference tree).  The arithmetic is written with an EXPLICIT operation order in
float32 (each numpy op rounds once, no fused multiply-add) so that the HIP
kernels, which are compiled with -ffp-contract=off and use the same order, can
be compared bit-for-bit on scores and argmax; against the reference's torch
implementation the agreement is to rounding (<=1e-4 relative), pinned by the
golden vectors under tests/golden/ that were generated from the reference
itself (tests/golden/make_golden.py).

Parity status: corners / projection / boxes / score_dimensions / propose
(deterministic part) are PINNED by golden vectors produced by the reference's
own code.  IoU (detectron2 `pairwise_iou`), the KD-tree chamfer
(`scipy.spatial.cKDTree`) and `cv2.minAreaRect` live in third-party code:
IoU is restated from detectron2's published definition and pinned only by
hand-computed cases ("parity unpinned" w.r.t. detectron2); the chamfer is
checked against scipy (installed); minAreaRect is an INPUT (rect_pts).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
F64 = np.float64


# --------------------------------------------------------------------------
# deterministic float32 exp: same algorithm (and op order) as the HIP kernel's
# cr_exp_f32, so dim scores agree bit-for-bit.  |rel err| vs libm <= 2 ulp.
# --------------------------------------------------------------------------
_LOG2E = F32(1.4426950408889634)
_LN2_HI = F32(0.693359375)            # 0x3f318000, exact in 9 bits
_LN2_LO = F32(-2.12194440e-4)
_EXP_C = [F32(1.9875691500e-4), F32(1.3981999507e-3), F32(8.3334519073e-3),
          F32(4.1665795894e-2), F32(1.6666665459e-1), F32(5.0000001201e-1)]


def exp_f32(x: np.ndarray) -> np.ndarray:
    """exp for x <= 0-ish arguments in pure float32 (Cephes expf scheme).
    x < -87 returns 0; NaN propagates."""
    x = np.asarray(x, dtype=F32)
    with np.errstate(invalid="ignore", over="ignore", under="ignore"):
        xc = np.maximum(x, F32(-87.0))      # NaN stays NaN (np.maximum propagates)
        xc = np.minimum(xc, F32(88.0))
        k = np.rint(xc * _LOG2E).astype(F32)
        r = xc - k * _LN2_HI
        r = r - k * _LN2_LO
        p = _EXP_C[0]
        for c in _EXP_C[1:]:
            p = p * r
            p = p + c
        r2 = r * r
        p = p * r2
        p = p + r
        p = p + F32(1.0)
        out = np.ldexp(p, k.astype(np.int32)).astype(F32)
        out = np.where(x < F32(-87.0), F32(0.0), out)
        out = np.where(np.isnan(x), F32(np.nan), out)
    return out.astype(F32)


# --------------------------------------------------------------------------
# corners -- cubercnn/util/math_util.py:142-245 (get_cuboid_verts_faces)
# --------------------------------------------------------------------------
# vertex sign table: x uses l, y uses h, z uses w  (math_util.py:198-207)
_SX = np.array([-1, 1, 1, -1, -1, 1, 1, -1], dtype=F32)   # x: -l/2 for {0,3,4,7}
_SY = np.array([-1, -1, 1, 1, -1, -1, 1, 1], dtype=F32)   # y: -h/2 for {0,1,4,5}
_SZ = np.array([-1, -1, -1, -1, 1, 1, 1, 1], dtype=F32)   # z: -w/2 for {0,1,2,3}


def cuboid_corners(box6: np.ndarray, R: np.ndarray) -> np.ndarray:
    """box6 (..., 6) = [X,Y,Z,W,H,L]; R (..., 3, 3) -> (..., 8, 3).
    verts = R @ local + centre; local = (+-l/2, +-h/2, +-w/2).
    math_util.py:186-219."""
    box6 = np.asarray(box6, dtype=F32)
    R = np.asarray(R, dtype=F32)
    hl = (box6[..., 5] / F32(2.0))[..., None]
    hh = (box6[..., 4] / F32(2.0))[..., None]
    hw = (box6[..., 3] / F32(2.0))[..., None]
    vx = _SX * hl            # (..., 8)   (-l/2 is exact negation)
    vy = _SY * hh
    vz = _SZ * hw
    out = np.empty(box6.shape[:-1] + (8, 3), dtype=F32)
    for a in range(3):
        acc = R[..., a, 0, None] * vx
        acc = acc + R[..., a, 1, None] * vy
        acc = acc + R[..., a, 2, None] * vz
        out[..., a] = acc + box6[..., a, None]
    return out


def cubes_corners(cubes: np.ndarray) -> np.ndarray:
    """Cubes.get_all_corners, ProposalNetwork/utils/spaces.py:192-204.
    cubes (N,P,15) -> (N,P,8,3)."""
    cubes = np.asarray(cubes, dtype=F32)
    return cuboid_corners(cubes[..., :6], cubes[..., 6:].reshape(cubes.shape[:-1] + (3, 3)))


def clamp_bounds(im_wh):
    """int(-c/2+1), int(c-1+c): python int() truncation toward zero,
    spaces.py:241-242."""
    c0, c1 = float(im_wh[0]), float(im_wh[1])
    return (F32(int(-c0 / 2 + 1)), F32(int(c0 - 1 + c0)),
            F32(int(-c1 / 2 + 1)), F32(int(c1 - 1 + c1)))


def project_corners(corners3d: np.ndarray, K: np.ndarray, im_wh=None) -> np.ndarray:
    """Cubes.get_bube_corners, spaces.py:224-245.  p = K @ X; u = p0/p2,
    v = p1/p2 with NO guard on p2<=0; optional clamp.  K is (3,3) or
    broadcastable (...,3,3) against corners3d (...,8,3)."""
    X = np.asarray(corners3d, dtype=F32)
    K = np.asarray(K, dtype=F32)
    if K.ndim == 2:
        Kb = K
        k = lambda i, j: Kb[i, j]
    else:
        k = lambda i, j: K[..., i, j, None]
    x, y, z = X[..., 0], X[..., 1], X[..., 2]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        p0 = (k(0, 0) * x + k(0, 1) * y) + k(0, 2) * z
        p1 = (k(1, 0) * x + k(1, 1) * y) + k(1, 2) * z
        p2 = (k(2, 0) * x + k(2, 1) * y) + k(2, 2) * z
        u = p0 / p2
        v = p1 / p2
    if im_wh is not None:
        lo0, hi0, lo1, hi1 = clamp_bounds(im_wh)
        # torch.clamp semantics: NaN stays NaN
        u = np.where(np.isnan(u), u, np.minimum(np.maximum(u, lo0), hi0))
        v = np.where(np.isnan(v), v, np.minimum(np.maximum(v, lo1), hi1))
    return np.stack((u, v), axis=-1).astype(F32)


def corners_to_boxes(corners2d: np.ndarray) -> np.ndarray:
    """cubes_to_box, ProposalNetwork/utils/conversions.py:25-48:
    [min u, min v, max u, max v] over the 8 corners.  torch.min/max propagate
    NaN, so does this."""
    c = np.asarray(corners2d, dtype=F32)
    with np.errstate(invalid="ignore"):
        return np.stack((c[..., 0].min(-1), c[..., 1].min(-1),
                         c[..., 0].max(-1), c[..., 1].max(-1)), axis=-1).astype(F32)


# --------------------------------------------------------------------------
# scores
# --------------------------------------------------------------------------
def iou_one_to_many(ref_box: np.ndarray, boxes: np.ndarray) -> np.ndarray:
    """score_iou -> iou_2d -> detectron2 pairwise_iou
    (scorefunction.py:47-49, utils.py:186-192).  detectron2's definition
    [3rd-party, restated]: wh = (min(rb) - max(lt)).clamp(0); inter = w*h;
    iou = inter / (a1 + a2 - inter) where inter > 0 else 0."""
    r = np.asarray(ref_box, dtype=F32).reshape(4)
    b = np.asarray(boxes, dtype=F32)
    with np.errstate(invalid="ignore", divide="ignore"):
        a1 = (r[2] - r[0]) * (r[3] - r[1])
        a2 = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
        w = np.minimum(r[2], b[..., 2]) - np.maximum(r[0], b[..., 0])
        h = np.minimum(r[3], b[..., 3]) - np.maximum(r[1], b[..., 1])
        # Tensor.clamp_(min=0) keeps NaN
        w = np.where(np.isnan(w), w, np.maximum(w, F32(0)))
        h = np.where(np.isnan(h), h, np.maximum(h, F32(0)))
        inter = w * h
        iou = inter / ((a1 + a2) - inter)
        return np.where(inter > 0, iou, F32(0)).astype(F32)


def score_dimensions(prior_mu, prior_sigma, dims, ref_box, boxes):
    """scorefunction.py:144-160.  dims (P,3) in (w,h,l); priors (3,) each.
    returns (score (P,), gauss (P,), diff (P,), maxdiff scalar)."""
    mu = np.asarray(prior_mu, dtype=F32).reshape(3)
    sg = np.asarray(prior_sigma, dtype=F32).reshape(3)
    d = np.asarray(dims, dtype=F32)
    r = np.asarray(ref_box, dtype=F32).reshape(4)
    b = np.asarray(boxes, dtype=F32)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        z = (d - mu) / sg
        e = exp_f32(F32(-0.5) * (z * z))          # -1/2 * z**2
        gauss = ((e[..., 0] + e[..., 1]) + e[..., 2]) / F32(3.0)
        gt_ratio = (r[2] - r[0]) / (r[3] - r[1])
        pr = (b[..., 2] - b[..., 0]) / (b[..., 3] - b[..., 1])
        diff = np.abs(gt_ratio - pr)
        maxdiff = _nanmax_torch(diff)
        score = (F32(1.0) - diff / maxdiff) * gauss
    return score.astype(F32), gauss.astype(F32), diff.astype(F32), F32(maxdiff)


def _nanmax_torch(x: np.ndarray):
    """torch.max over a vector: NaN if any NaN, else max."""
    x = np.asarray(x, dtype=F32)
    if x.size == 0:
        return F32(np.nan)
    if np.isnan(x).any():
        return F32(np.nan)
    return F32(x.max())


def corner_chamfer(rect_pts: np.ndarray, corners2d: np.ndarray) -> np.ndarray:
    """modified_chamfer_distance over all proposals, scorefunction.py:51-56,79-81:
    for each of the 4 rect points the min Euclidean distance to the 8 projected
    corners, in float64 (scipy cKDTree works in double), mean of the 4 (np.mean
    in float64), then stored to a float32 torch tensor.  -> (P,) float32."""
    rp = np.asarray(rect_pts, dtype=F32).astype(F64).reshape(4, 2)
    c = np.asarray(corners2d, dtype=F32).astype(F64)            # (P,8,2)
    dx = rp[None, :, None, 0] - c[:, None, :, 0]
    dy = rp[None, :, None, 1] - c[:, None, :, 1]
    with np.errstate(invalid="ignore", over="ignore"):
        d2 = dx * dx + dy * dy                                       # (P,4,8)
        # NaN corner -> cKDTree would misbehave; we define NaN-propagating min
        m = np.where(np.isnan(d2).any(-1), np.nan, d2.min(-1))
        d = np.sqrt(m)
        s = (((d[:, 0] + d[:, 1]) + d[:, 2]) + d[:, 3]) / F64(4.0)
    return s.astype(F32)


def fallback_rect(corners2d: np.ndarray) -> np.ndarray:
    """score_corners' fallback when the mask has no contour,
    scorefunction.py:69-75: axis-aligned box from the mean over proposals of the
    per-proposal min/max u,v.  (float32 mean; torch's summation order is not
    restated -- tolerance, not bit-exact.)"""
    c = np.asarray(corners2d, dtype=F32)
    mnx = F32(np.mean(c[..., 0].min(-1), dtype=F64))
    mxx = F32(np.mean(c[..., 0].max(-1), dtype=F64))
    mny = F32(np.mean(c[..., 1].min(-1), dtype=F64))
    mxy = F32(np.mean(c[..., 1].max(-1), dtype=F64))
    return np.array([[mnx, mny], [mxx, mny], [mxx, mxy], [mnx, mxy]], dtype=F32)


def score_corners_from_rect(rect_pts, corners2d):
    """1 - s / max_P s, scorefunction.py:83-85."""
    s = corner_chamfer(rect_pts, corners2d)
    with np.errstate(invalid="ignore", divide="ignore"):
        mx = _nanmax_torch(s)
        return (F32(1.0) - s / mx).astype(F32), s, mx


def argmax_numpy(x: np.ndarray) -> int:
    """np.argmax semantics (roi_heads.py:502): first maximal index, NaN counts
    as maximal (first NaN wins)."""
    return int(np.argmax(np.asarray(x, dtype=F32)))


def project_and_score(cubes, K, im_wh, ref_boxes, prior_mu, prior_sigma, rect_pts=None, iou_boxes=None):
    """The AP path of ROIHeads_Boxer._forward_cube, roi_heads.py:492-505, for N
    objects x P proposals, one object at a time like the reference's loop.

    cubes (N,P,15); K (3,3) or (N,3,3); im_wh (W,H); ref_boxes (N,4);
    prior_mu/prior_sigma (N,3); rect_pts (N,4,2) or None (=> fallback rect).
    Returns dict of corners (N,P,8,2), boxes (N,P,4), iou/dim/corner/combined
    (N,P), argmax (N,) int64, best (N,)."""
    cubes = np.asarray(cubes, dtype=F32)
    N, P = cubes.shape[:2]
    K = np.asarray(K, dtype=F32)
    out = dict(corners=np.zeros((N, P, 8, 2), F32), boxes=np.zeros((N, P, 4), F32),
               iou=np.zeros((N, P), F32), dim=np.zeros((N, P), F32),
               corner=np.zeros((N, P), F32), combined=np.zeros((N, P), F32),
               argmax=np.zeros((N,), np.int64), best=np.zeros((N,), F32))
    for i in range(N):
        Ki = K if K.ndim == 2 else K[i]
        c3 = cubes_corners(cubes[i])
        c2 = project_corners(c3, Ki, im_wh)
        bx = corners_to_boxes(c2)
        # (the GT-box branches score IoU against the projected ground-truth cube: roi_heads.py:459,530)
        iou = iou_one_to_many(ref_boxes[i] if iou_boxes is None else iou_boxes[i], bx)
        dim, _, _, _ = score_dimensions(prior_mu[i], prior_sigma[i], cubes[i, :, 3:6], ref_boxes[i], bx)
        rect = fallback_rect(c2) if rect_pts is None else rect_pts[i]
        cor, _, _ = score_corners_from_rect(rect, c2)
        with np.errstate(invalid="ignore"):
            comb = (iou * dim) * cor           # roi_heads.py:499 order
        out["corners"][i] = c2
        out["boxes"][i] = bx
        out["iou"][i] = iou
        out["dim"][i] = dim
        out["corner"][i] = cor
        out["combined"][i] = comb
        if P:
            a = argmax_numpy(comb)
            out["argmax"][i] = a
            out["best"][i] = comb[a]
    return out


# --------------------------------------------------------------------------
# proposal sampler (deterministic given the random draws)
# --------------------------------------------------------------------------
def vec_perp(n: np.ndarray) -> np.ndarray:
    """vec_perp_t, utils.py:112-118: [0,c,-b] (un-normalised) if a == 0 else
    normalize([b,-a,0]) with the 1e-8 floor of normalize_vector (utils.py:10-16)."""
    a, b, c = [F32(v) for v in np.asarray(n, dtype=F32)]
    if a == 0:
        return np.array([0, c, -b], dtype=F32)
    v = np.array([b, -a, 0], dtype=F32)
    mag = np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2], dtype=F32)
    mag = np.maximum(mag, F32(1e-8))
    return (v / mag).astype(F32)


def cross3(a, b):
    a = np.asarray(a, dtype=F32)
    b = np.asarray(b, dtype=F32)
    return np.stack((a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]), axis=-1).astype(F32)


def yaw_table(normal: np.ndarray, n_yaw: int = 36) -> np.ndarray:
    """orthobasis_from_normal_t(normal, linspace(0,pi,36)), utils.py:120-146,
    proposals.py:404-406: x_t = p cos t + (n x p) sin t + n (n.p)(1-cos t);
    y_t = n x x_t; COLUMNS of R are (x_t, n, y_t).  -> (n_yaw,3,3)."""
    n = np.asarray(normal, dtype=F32).reshape(3)
    # torch.linspace(0, pi, 36) in float32: start + i*step with step=(end-start)/(steps-1),
    # symmetric fill from both ends for i >= steps/2 (ATen linspace kernel)
    end = F32(np.pi)
    step = end / F32(n_yaw - 1)
    idx = np.arange(n_yaw)
    th = np.where(idx < n_yaw // 2, F32(0) + step * idx.astype(F32),
                  end - step * (n_yaw - 1 - idx).astype(F32)).astype(F32)
    p = vec_perp(n)
    ct = np.cos(th).astype(F32)[:, None]
    st = np.sin(th).astype(F32)[:, None]
    kxp = cross3(n, p)[None, :]
    kdotp = F32((n[0] * p[0] + n[1] * p[1]) + n[2] * p[2])
    x = (p[None, :] * ct + kxp * st) + (n * kdotp)[None, :] * (F32(1) - ct)
    y = cross3(np.broadcast_to(n, x.shape), x)
    R = np.stack((x, np.broadcast_to(n, x.shape), y), axis=-1)    # columns
    return R.astype(F32)


def propose_from_draws(ref_boxes, depth, prior_mu, prior_sigma, K, P,
                       dim_normals, ctr_normals, yaw_idx, normal):
    """Deterministic restatement of proposals.propose, proposals.py:338-424,
    with the random variates supplied by the caller:

      dim_normals (R,3,N,P) standard normals: round r of the rejection sampler of
          sample_normal_in_range (utils.py:42-60) for w,h,l; an entry is only
          consumed if the previous rounds left it invalid.
      ctr_normals (3,N,P) standard normals for x,y,z.
      yaw_idx (N,P) int in [0,36).

    sample = mean + std * n  (what torch.normal(mean,std) computes from its
    standard-normal draw).  Returns cubes (N,P,15) float32."""
    b = np.asarray(ref_boxes, dtype=F32)
    depth = np.asarray(depth, dtype=F32)
    K = np.asarray(K, dtype=F32)
    mu = np.asarray(prior_mu, dtype=F32)
    sg = np.asarray(prior_sigma, dtype=F32)
    N = b.shape[0]
    m = F32(4)
    widths = b[:, 2] - b[:, 0]
    heights = b[:, 3] - b[:, 1]
    x_lo, x_hi = b[:, 0] + widths / m, b[:, 2] - widths / m
    y_lo, y_hi = b[:, 1] + heights / m, b[:, 3] - heights / m
    ar = np.arange(P, dtype=F32)[None, :]
    # vectorized_linspace, utils.py:170-177, then .long() (trunc toward zero)
    xg = np.trunc(ar * ((x_hi - x_lo) / F32(P - 1))[:, None] + x_lo[:, None]).astype(np.int64)
    yg = np.trunc(ar * ((y_hi - y_lo) / F32(P - 1))[:, None] + y_lo[:, None]).astype(np.int64)
    d = depth[yg, xg]                                   # diagonal samples (N,P)
    ox = xg.astype(F32) - K[0, 2]
    oy = yg.astype(F32) - K[1, 2]
    a = K[0, 0]
    ang_x = np.arctan2(ox, a).astype(F32)
    dxc = np.sqrt(ox * ox + a * a).astype(F32)
    ang_d = np.arctan2(oy, dxc).astype(F32)
    y = d * np.sin(ang_d).astype(F32)
    with np.errstate(invalid="ignore"):
        dx = np.sqrt(d * d - y * y).astype(F32)
        x = dx * np.sin(ang_x).astype(F32)
        z_tmp = np.sqrt(dx * dx - x * x).astype(F32)

    def trunc_normal(mean, std, lo, hi, draws):
        s = mean[:, None] + std[:, None] * draws[0]
        for r in range(1, draws.shape[0]):
            bad = (s < lo) | (s > hi[:, None])
            if not bad.any():
                break
            s = np.where(bad, mean[:, None] + std[:, None] * draws[r], s)
        return s.astype(F32)

    lo = F32(0.05)
    w = trunc_normal(mu[:, 0], sg[:, 0], lo, mu[:, 0] + F32(2) * sg[:, 0], dim_normals[:, 0])
    h = trunc_normal(mu[:, 1], sg[:, 1] * F32(1.1), lo, mu[:, 1] + F32(2.2) * sg[:, 1], dim_normals[:, 1])
    l = trunc_normal(mu[:, 2], sg[:, 2], lo, mu[:, 2] + F32(2) * sg[:, 2], dim_normals[:, 2])

    def lower_median(v):       # torch.median: lower of the two middle values
        return np.sort(v, axis=1)[:, (v.shape[1] - 1) // 2]

    def std_unbiased(v):
        return np.std(v.astype(F64), axis=1, ddof=1).astype(F32)

    xs = (F32(1.15) * lower_median(x) + F32(0))[:, None] + (std_unbiased(x) * F32(1.2))[:, None] * ctr_normals[0]
    ys = (F32(1.1) * lower_median(y) + F32(0))[:, None] + (std_unbiased(y) * F32(0.8))[:, None] * ctr_normals[1]
    zz = z_tmp + l / F32(2)
    zs = (F32(0.85) * lower_median(zz) + F32(0.35))[:, None] + (std_unbiased(zz) * F32(1.2))[:, None] * ctr_normals[2]
    Rt = yaw_table(normal)[np.asarray(yaw_idx)]           # (N,P,3,3)
    cubes = np.concatenate((np.stack((xs, ys, zs, w, h, l), axis=2).astype(F32),
                            Rt.reshape(N, P, 9)), axis=2)
    return cubes.astype(F32)


# --------------------------------------------------------------------------
# RANSAC ground plane (given the sampled triples)
# --------------------------------------------------------------------------
def ransac_plane(pts, triples, thresh=0.05):
    """Plane.fit_parallel, ProposalNetwork/utils/plane.py:79-134, with the
    `random.sample` triples given.  Returns (-equation (4,), inlier count,
    best index).  argmax = first max (torch.argmax)."""
    pts = np.asarray(pts, dtype=F32)
    tr = np.asarray(triples)
    p0, p1, p2 = pts[tr[:, 0]], pts[tr[:, 1]], pts[tr[:, 2]]
    vA, vB = p1 - p0, p2 - p0
    vC = cross3(vA, vB)
    with np.errstate(invalid="ignore", divide="ignore"):
        nrm = np.sqrt((vC[:, 0] * vC[:, 0] + vC[:, 1] * vC[:, 1]) + vC[:, 2] * vC[:, 2]).astype(F32)
        vC = vC / nrm[:, None]
        k = -((vC[:, 0] * p1[:, 0] + vC[:, 1] * p1[:, 1]) + vC[:, 2] * p1[:, 2])
        den = np.sqrt((vC[:, 0] ** 2 + vC[:, 1] ** 2) + vC[:, 2] ** 2).astype(F32)
        dist = (((vC[:, 0, None] * pts[None, :, 0] + vC[:, 1, None] * pts[None, :, 1])
                 + vC[:, 2, None] * pts[None, :, 2]) + k[:, None]) / den[:, None]
        inl = np.abs(dist) <= F32(thresh)
    counts = inl.sum(1)
    best = int(np.argmax(counts))
    eq = np.array([vC[best, 0], vC[best, 1], vC[best, 2], k[best]], dtype=F32)
    return -eq, int(counts[best]), best, counts.astype(np.int32)


def fix_ground_normal(nv):
    """roi_heads.py:411-428 axis fix-ups (numpy float64 in the reference)."""
    nv = np.asarray(nv, dtype=F64).copy()
    if abs(nv[2]) > abs(nv[1]):
        nv = np.array([nv[0], nv[2], -nv[1]])
    if abs(nv[0]) > abs(nv[1]):
        nv = np.array([-nv[2], nv[0], nv[1]])
    if nv[1] < 0:
        nv = nv * -1
    return nv


# --------------------------------------------------------------------------
# mask scores (MABO): hull raster counts on the [::stride, ::stride] grid
# --------------------------------------------------------------------------
def segment_counts(corners2d, mask, stride=4):
    """restatement of cv2.convexHull -> int32 -> cv2.fillPoly -> [::stride, ::stride] used by score_segmentation
    (ProposalNetwork/scoring/scorefunction.py:88-105) [cv2: third-party, absent]: a grid sample belongs to the polygon
    iff it lies inside or on the closed convex hull of the 8 points, hull vertices truncated to int.  -> (P,2) int64
    {polygon samples, polygon AND mask samples}."""
    c = np.asarray(corners2d, dtype=F32)
    m = np.asarray(mask) != 0
    H, W = m.shape
    out = np.zeros((c.shape[0], 2), dtype=np.int64)
    ys, xs = np.meshgrid(np.arange(0, H, stride), np.arange(0, W, stride), indexing="ij")
    for p in range(c.shape[0]):
        pts = c[p]
        if not np.isfinite(pts).all() or np.abs(pts).max() >= 1e9:
            continue
        hull = _hull_ccw(pts)
        hv = np.array([[int(x), int(y)] for x, y in hull], dtype=np.int64)          # int(): truncation toward zero
        k = len(hv)
        if k == 1:
            inside = (xs == hv[0, 0]) & (ys == hv[0, 1])
        else:
            pos = np.ones_like(xs, dtype=bool)
            neg = np.ones_like(xs, dtype=bool)
            for e in range(k):
                a, b = hv[e], hv[(e + 1) % k]
                cr = (b[0] - a[0]) * (ys - a[1]) - (b[1] - a[1]) * (xs - a[0])
                pos &= cr >= 0
                neg &= cr <= 0
            inside = pos | neg
            if k == 2:
                inside &= (xs >= hv[:, 0].min()) & (xs <= hv[:, 0].max()) & (ys >= hv[:, 1].min()) & (ys <= hv[:, 1].max())
        out[p, 0] = inside.sum()
        out[p, 1] = (inside & m[::stride, ::stride]).sum()
    return out


def _hull_ccw(pts):
    """convex hull vertices of a few float32 points by gift wrapping with exact float32 cross products (collinear:
    farthest), starting at the lowest x (then lowest y)"""
    pts = np.asarray(pts, dtype=F32)
    n = len(pts)
    start = 0
    for i in range(1, n):
        if pts[i, 0] < pts[start, 0] or (pts[i, 0] == pts[start, 0] and pts[i, 1] < pts[start, 1]):
            start = i
    hull, l = [], start
    for _ in range(n):
        hull.append(pts[l])
        q = -1
        for i in range(n):
            if pts[i, 0] == pts[l, 0] and pts[i, 1] == pts[l, 1]:
                continue
            if q < 0:
                q = i
                continue
            d = F32(F32(pts[i, 0] - pts[l, 0]) * F32(pts[q, 1] - pts[l, 1])) - F32(F32(pts[i, 1] - pts[l, 1]) * F32(pts[q, 0] - pts[l, 0]))
            di = F32(F32(pts[i, 0] - pts[l, 0]) ** 2) + F32(F32(pts[i, 1] - pts[l, 1]) ** 2)
            dq = F32(F32(pts[q, 0] - pts[l, 0]) ** 2) + F32(F32(pts[q, 1] - pts[l, 1]) ** 2)
            if d > 0 or (d == 0 and di > dq):
                q = i
        if q < 0 or (pts[q, 0] == pts[start, 0] and pts[q, 1] == pts[start, 1]):
            break
        l = q
    return hull
