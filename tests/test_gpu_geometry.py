"""GPU parity: the HIP geometry kernels (through the C-ABI) against the numpy
oracle (bit-exact scores/argmax: same float32 op order) and against the golden
vectors generated from the reference (1e-4 relative, argmax exact)."""
import importlib
import os

import numpy as np
import pytest
import torch

from oracle import geometry as og

pytestmark = pytest.mark.gpu
geo = importlib.import_module("3dod_amd.geometry")
DEV = "cuda:0"


def T(x, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x), dtype=dtype, device=DEV)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def same_bits(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return ((a == b) | (np.isnan(a) & np.isnan(b))).all()


def assert_close(a, b, rtol=1e-4, atol=1e-5):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape
    assert (np.isnan(a) == np.isnan(b)).all()
    m = np.isfinite(a) & np.isfinite(b)
    assert (a[~m & ~np.isnan(a)] == b[~m & ~np.isnan(b)]).all()
    np.testing.assert_allclose(a[m], b[m], rtol=rtol, atol=atol)


def run_k17(cubes, K, im_wh, ref, mu, sg, rect):
    out = geo.cubes_project_score(T(cubes), T(K), im_wh, T(ref), T(mu), T(sg), None if rect is None else T(rect))
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items() if v is not None}


def random_case(N, P, seed, nasty=True):
    rng = np.random.default_rng(seed)
    K = np.array([[rng.uniform(400, 800), 0, 256], [0, rng.uniform(400, 800), 256], [0, 0, 1]], np.float32)
    z = rng.uniform(1, 8, (N, P)); u = rng.uniform(0, 512, (N, P)); v = rng.uniform(0, 512, (N, P))
    x = (u - 256) * z / K[0, 0]; y = (v - 256) * z / K[1, 1]
    dims = rng.uniform(0.05, 1.5, (N, P, 3))
    q = rng.normal(size=(N, P, 4)); q /= np.linalg.norm(q, axis=-1, keepdims=True)
    w_, x_, y_, z_ = np.moveaxis(q, -1, 0)
    R = np.stack([1 - 2 * (y_**2 + z_**2), 2 * (x_ * y_ - z_ * w_), 2 * (x_ * z_ + y_ * w_),
                  2 * (x_ * y_ + z_ * w_), 1 - 2 * (x_**2 + z_**2), 2 * (y_ * z_ - x_ * w_),
                  2 * (x_ * z_ - y_ * w_), 2 * (y_ * z_ + x_ * w_), 1 - 2 * (x_**2 + y_**2)], -1)
    cubes = np.concatenate([x[..., None], y[..., None], z[..., None], dims, R], -1).astype(np.float32)
    if nasty and P >= 8:
        cubes[:, 0, 2] = 0.01          # almost at the camera
        cubes[:, 1, 2] = -1.0          # behind the camera (unguarded divide)
        cubes[:, 2, 0] = 80.0          # far off-screen -> clamp
    ctr = rng.uniform(100, 400, (N, 2)); wh = rng.uniform(40, 190, (N, 2))
    ref = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
    mu = rng.uniform(0.3, 1.1, (N, 3)).astype(np.float32); sg = (0.2 * mu).astype(np.float32)
    rect = (np.stack([ref[:, [0, 1]], ref[:, [2, 1]], ref[:, [2, 3]], ref[:, [0, 3]]], 1)
            + rng.normal(0, 3, (N, 4, 2))).astype(np.float32)
    return cubes, K, (512, 512), ref, mu, sg, rect


def test_g1_corners(golden_dir):
    g = load(golden_dir, "geometry_g1_corners.npz")
    v = geo.cuboid_corners(T(g["box6"]), T(g["R"])).cpu().numpy()
    assert same_bits(v, og.cuboid_corners(g["box6"], g["R"]))
    assert_close(v, g["verts"], atol=1e-5)


def test_g2_golden(golden_dir):
    g = load(golden_dir, "geometry_g2_project_score.npz")
    out = run_k17(g["cubes"], g["K"], tuple(g["im_wh"]), g["ref_boxes"], g["prior_mu"], g["prior_sigma"], g["rect_pts"])
    # vs the reference's own outputs
    frac = np.mean(np.abs(out["corners"] - g["corners2d"]) <= 1e-4 * np.abs(g["corners2d"]) + 1e-3)
    assert frac > 0.999
    assert_close(out["corners"], g["corners2d"], rtol=1e-4, atol=2e-2)
    assert_close(out["iou"], g["iou"], atol=2e-5)
    assert_close(out["dim"], g["dim"], atol=2e-5)
    assert_close(out["corner"], g["corner"], atol=2e-5)
    assert_close(out["combined"], g["combined"], atol=1e-5)
    assert (out["argmax"] == g["argmax"]).all()          # bit-exact index
    # vs the oracle: bit-for-bit
    o = og.project_and_score(g["cubes"], g["K"], tuple(g["im_wh"]), g["ref_boxes"], g["prior_mu"],
                             g["prior_sigma"], g["rect_pts"])
    for k in ("corners", "boxes", "iou", "dim", "corner", "combined"):
        assert same_bits(out[k], o[k]), k
    assert (out["argmax"] == o["argmax"]).all()
    assert same_bits(out["best"], o["best"])


@pytest.mark.parametrize("N,P", [(1, 1), (2, 7), (3, 255), (3, 256), (2, 257), (5, 1000), (2, 1024), (2, 1025),
                                 (1, 4096), (17, 33)])
def test_random_bitexact_vs_oracle(N, P):
    case = random_case(N, P, seed=N * 10007 + P)
    out = run_k17(*case)
    o = og.project_and_score(*case)
    for k in ("corners", "boxes", "iou", "dim", "corner", "combined"):
        assert same_bits(out[k], o[k]), (k, N, P)
    assert (out["argmax"] == o["argmax"]).all()


def test_k_per_object_and_argmax_only():
    cubes, K, im, ref, mu, sg, rect = random_case(6, 1000, seed=3)
    Ks = np.stack([K * np.array([[1 + 0.05 * i, 1, 1], [1, 1 + 0.03 * i, 1], [1, 1, 1]], np.float32) for i in range(6)])
    full = geo.cubes_project_score(T(cubes), T(Ks), im, T(ref), T(mu), T(sg), T(rect))
    lean = geo.cubes_project_score(T(cubes), T(Ks), im, T(ref), T(mu), T(sg), T(rect), want=())
    o = og.project_and_score(cubes, Ks, im, ref, mu, sg, rect)
    assert (full["argmax"].cpu().numpy() == o["argmax"]).all()
    assert (lean["argmax"].cpu().numpy() == o["argmax"]).all()
    assert same_bits(lean["best"].cpu().numpy(), o["best"])
    assert lean["corners"] is None


def test_separate_iou_box_bitexact_vs_oracle():
    """iou_boxes: the IoU term against another box than the aspect-ratio / corner terms (the GT-box branches of
    ROIHeads_Boxer score IoU2D against the projected ground-truth cube, roi_heads.py:459,530)"""
    cubes, K, im, ref, mu, sg, rect = random_case(5, 1000, seed=11)
    rng = np.random.default_rng(4)
    iou_ref = (ref + rng.normal(0, 12, ref.shape)).astype(np.float32)
    out = geo.cubes_project_score(T(cubes), T(K), im, T(ref), T(mu), T(sg), T(rect), iou_boxes=T(iou_ref))
    o = og.project_and_score(cubes, K, im, ref, mu, sg, rect, iou_boxes=iou_ref)
    o0 = og.project_and_score(cubes, K, im, ref, mu, sg, rect)
    for k in ("corners", "boxes", "iou", "dim", "corner", "combined"):
        assert same_bits(out[k].cpu().numpy(), o[k]), k
    assert (out["argmax"].cpu().numpy() == o["argmax"]).all()
    assert same_bits(o["dim"], o0["dim"]) and not same_bits(o["iou"], o0["iou"])      # only the IoU term moved


def test_nan_semantics():
    # zero-height projected boxes -> 0/0 ratio -> NaN dim score; np.argmax picks the first NaN
    cubes, K, im, ref, mu, sg, rect = random_case(2, 64, seed=5, nasty=False)
    cubes[0, 10, 3:6] = 0.0            # degenerate cube: all 8 corners coincide -> zero-size box
    cubes[0, 30, 3:6] = 0.0
    out = run_k17(cubes, K, im, ref, mu, sg, rect)
    o = og.project_and_score(cubes, K, im, ref, mu, sg, rect)
    assert np.isnan(o["combined"][0]).any()
    for k in ("dim", "combined"):
        assert same_bits(out[k], o[k])
    assert (out["argmax"] == o["argmax"]).all()


def test_fallback_rect():
    cubes, K, im, ref, mu, sg, _ = random_case(4, 1000, seed=9, nasty=False)
    out = run_k17(cubes, K, im, ref, mu, sg, None)
    o = og.project_and_score(cubes, K, im, ref, mu, sg, None)
    assert_close(out["corner"], o["corner"], rtol=1e-4, atol=1e-4)     # mean over P: summation order differs
    assert same_bits(out["iou"], o["iou"])


def test_empty_and_errors():
    lib_mod = importlib.import_module("3dod_amd._lib")
    e = geo.cubes_project_score(torch.zeros(0, 1000, 15, device=DEV), torch.eye(3, device=DEV), (512, 512),
                                torch.zeros(0, 4, device=DEV), torch.zeros(0, 3, device=DEV), torch.zeros(0, 3, device=DEV))
    assert e["argmax"].shape == (0,)
    with pytest.raises(lib_mod.CrError):
        geo.cubes_project_score(torch.zeros(1, 5000, 15, device=DEV), torch.eye(3, device=DEV), (512, 512),
                                torch.zeros(1, 4, device=DEV), torch.ones(1, 3, device=DEV), torch.ones(1, 3, device=DEV))
    with pytest.raises(TypeError):
        geo.cuboid_corners(torch.zeros(2, 6, device=DEV, dtype=torch.float64), torch.zeros(2, 3, 3, device=DEV))


def test_full_size_properties():
    """BASELINE config 3: 64 images x 16 objects x 1000 cubes in ONE launch."""
    N, P = 1024, 1000
    cubes, K, im, ref, mu, sg, rect = random_case(N, P, seed=11)
    a = geo.cubes_project_score(T(cubes), T(K), im, T(ref), T(mu), T(sg), T(rect))
    b = geo.cubes_project_score(T(cubes), T(K), im, T(ref), T(mu), T(sg), T(rect))
    torch.cuda.synchronize()
    comb = a["combined"].cpu().numpy(); am = a["argmax"].cpu().numpy(); best = a["best"].cpu().numpy()
    assert ((am >= 0) & (am < P)).all()
    assert same_bits(comb[np.arange(N), am], best)                 # best == combined[argmax]
    assert (am == np.array([og.argmax_numpy(c) for c in comb])).all()      # argmax of its own plane, numpy rule
    assert same_bits(comb, b["combined"].cpu().numpy()) and (am == b["argmax"].cpu().numpy()).all()   # deterministic
    bx = a["boxes"].cpu().numpy(); c2 = a["corners"].cpu().numpy()
    assert same_bits(bx, og.corners_to_boxes(c2))                   # boxes are the min/max of the written corners
    iou = a["iou"].cpu().numpy(); assert ((iou >= 0) & (iou <= 1)).all()
    sub = np.arange(0, N, 64)
    o = og.project_and_score(cubes[sub], K, im, ref[sub], mu[sub], sg[sub], rect[sub])
    assert same_bits(comb[sub], o["combined"]) and (am[sub] == o["argmax"]).all()


def test_propose_golden(golden_dir):
    g = load(golden_dir, "geometry_g6_propose.npz")
    normals = g["normals"]; N = g["boxes"].shape[0]; P = int(g["P"])
    mu, sg = g["prior_mu"], g["prior_sigma"]
    ctr = normals[-3:]; dd = normals[:-3]
    f = np.float32

    def rounds(mean, std, hi, draws):
        s = mean[:, None] + std[:, None] * draws[0]; r = 1
        while ((s < f(0.05)) | (s > hi[:, None])).any():
            bad = (s < f(0.05)) | (s > hi[:, None])
            s = np.where(bad, mean[:, None] + std[:, None] * draws[r], s); r += 1
        return r
    rw = rounds(mu[:, 0], sg[:, 0], mu[:, 0] + f(2) * sg[:, 0], dd)
    rh = rounds(mu[:, 1], sg[:, 1] * f(1.1), mu[:, 1] + f(2.2) * sg[:, 1], dd[rw:])
    rl = dd.shape[0] - rw - rh
    Rn = max(rw, rh, rl)
    dn = np.zeros((Rn, 3, N, P), np.float32)
    dn[:rw, 0] = dd[:rw]; dn[:rh, 1] = dd[rw:rw + rh]; dn[:rl, 2] = dd[rw + rh:]
    cubes, exhausted = geo.propose_from_draws(T(g["boxes"]), T(g["depth"]), T(mu), T(sg), T(g["K"]), P, T(dn), T(ctr),
                                              T(g["yaw_idx"], torch.int32), T(g["normal"]))
    assert int(exhausted.item()) == 0
    c = cubes.cpu().numpy()
    assert_close(c[..., 3:], g["cubes"][..., 3:], atol=3e-6)
    assert_close(c[..., :3], g["cubes"][..., :3], rtol=2e-4, atol=2e-4)
    o = og.propose_from_draws(g["boxes"], g["depth"], mu, sg, g["K"], P, dn, ctr, g["yaw_idx"], g["normal"])
    assert_close(c, o, rtol=2e-4, atol=2e-4)
    # too few rounds is reported, not hidden
    _, ex2 = geo.propose_from_draws(T(g["boxes"]), T(g["depth"]), T(mu), T(sg), T(g["K"]), P, T(dn[:1]), T(ctr),
                                    T(g["yaw_idx"], torch.int32), T(g["normal"]))
    assert int(ex2.item()) > 0


def test_ransac_golden(golden_dir):
    g = load(golden_dir, "geometry_g8_ransac.npz")
    neg_eq, counts, best = geo.ransac_plane(T(g["pts"]), T(g["triples"], torch.int32), float(g["thresh"]))
    o_eq, o_cnt, o_best, o_counts = og.ransac_plane(g["pts"], g["triples"], float(g["thresh"]))
    assert int(best[1].item()) == int(g["n_inliers"]) == o_cnt
    assert int(best[0].item()) == o_best
    assert (counts.cpu().numpy() == o_counts).all()
    assert_close(neg_eq.cpu().numpy(), g["neg_equation"], atol=1e-6)
    with pytest.raises(ValueError):
        geo.ransac_plane(T(g["pts"]), T(np.array([[0, 1, 10**6]]), torch.int32))


# ---------------------------------------------------------------------------
# cr_cubes_project_score_fast: argmax / best bit-equal to the exact kernel (and so to the oracle), planes to 1e-4
# ---------------------------------------------------------------------------
PLANES = ("corners", "boxes", "iou", "dim", "corner", "combined")


def both(case, want=PLANES, **kw):
    cubes, K, im, ref, mu, sg, rect = case
    args = (T(cubes), T(K), im, T(ref), T(mu), T(sg), None if rect is None else T(rect))
    st = torch.zeros(2, dtype=torch.int64, device=DEV)
    ex = geo.cubes_project_score(*args, want=want, fast=False, **kw)
    fa = geo.cubes_project_score(*args, want=want, fast=True, stats=st, **kw)
    torch.cuda.synchronize()
    n = lambda d: {k: v.cpu().numpy() for k, v in d.items() if v is not None}
    return n(ex), n(fa), st.cpu().numpy()


def check_fast(ex, fa):
    assert (fa["argmax"] == ex["argmax"]).all()
    assert same_bits(fa["best"], ex["best"])
    for k in PLANES:
        if k in ex:
            assert_close(fa[k], ex[k], rtol=1e-4, atol=2e-4 if k in ("corners", "boxes") else 1e-5)


@pytest.mark.parametrize("N,P", [(1, 1), (2, 7), (3, 255), (3, 256), (2, 257), (64, 1000), (2, 1024), (2, 1025), (1, 4096)])
def test_fast_clean_inputs(N, P):
    """no cube near the camera plane: the objects go through the fast planes + candidate re-evaluation"""
    case = random_case(N, P, seed=N * 313 + P, nasty=False)
    ex, fa, st = both(case)
    check_fast(ex, fa)
    o = og.project_and_score(*case)
    assert (fa["argmax"] == o["argmax"]).all() and same_bits(fa["best"], o["best"])
    if P >= 255:
        assert st[0] < N, "every object fell back to the exact sequence"
        assert st[1] <= 64 * N


def test_fast_error_is_far_inside_the_candidate_intervals():
    """the measured fast-vs-exact differences on BASELINE configs[2]-shaped input against the interval half-widths the
    kernel uses (4x the analytic bound): chamfer 4*delta px, combined 4e-3 relative"""
    case = random_case(256, 1000, seed=99, nasty=False)
    ex, fa, st = both(case)
    check_fast(ex, fa)
    delta = 1023.0 / 4194304.0
    m = ex["combined"].max(axis=1, keepdims=True)
    rel = np.abs(fa["combined"] - ex["combined"]) / np.maximum(m, 1e-6)
    assert rel.max() < 4e-3 / 8, rel.max()
    assert np.abs(fa["boxes"] - ex["boxes"]).max() <= delta
    assert np.abs(fa["corner"] - ex["corner"]).max() < 1e-5


def test_fast_nasty_inputs():
    """cubes almost at / behind the camera and far off-screen (finite values: fast path), and cubes with a corner exactly
    on the camera plane (p2 = 0: inf / NaN coordinates -> those objects run the exact sequence, planes bit-equal)"""
    case = random_case(6, 1000, seed=5, nasty=True)
    ex, fa, st = both(case)
    check_fast(ex, fa)
    o = og.project_and_score(*case)
    assert (fa["argmax"] == o["argmax"]).all() and same_bits(fa["best"], o["best"])
    cubes = case[0].copy()
    for n in (1, 4):
        cubes[n, 7, :3] = [0.3, -0.2, 0.5]                   # centre z 0.5, extent 1.0 along z, identity pose:
        cubes[n, 7, 3:6] = [1.0, 0.6, 0.8]                   # four corners at z = 0 exactly
        cubes[n, 7, 6:] = np.eye(3, dtype=np.float32).ravel()
    case = (cubes,) + case[1:]
    ex, fa, st = both(case)
    assert st[0] == 2
    for k in PLANES:
        assert same_bits(fa[k][[1, 4]], ex[k][[1, 4]]), k
    check_fast(ex, fa)
    o = og.project_and_score(*case)
    assert (fa["argmax"] == o["argmax"]).all() and same_bits(fa["best"], o["best"])


def test_fast_ties_empty_overlap_no_rect_and_iou_boxes():
    cubes, K, im, ref, mu, sg, rect = random_case(8, 1000, seed=21, nasty=False)
    cubes[:, 500:] = cubes[:, :500]                          # every cube has a twin: first index wins
    ref[3] = [-2000.0, -2000.0, -1990.0, -1985.0]            # outside the clamp range: every combined score is 0 -> index 0
    rect[5] = np.nan                                         # empty mask: the fallback rectangle
    case = (cubes, K, im, ref, mu, sg, rect)
    ex, fa, st = both(case)
    check_fast(ex, fa)
    assert (fa["argmax"] < 500).all() and fa["argmax"][3] == 0
    assert st[0] >= 2
    o = og.project_and_score(*case)
    keep = np.arange(8) != 5                                 # (a NaN row means "fallback rectangle" to the kernels only)
    assert (fa["argmax"][keep] == o["argmax"][keep]).all() and same_bits(fa["best"][keep], o["best"][keep])
    o5 = og.project_and_score(cubes[5:6], K, im, ref[5:6], mu[5:6], sg[5:6], None)
    assert fa["argmax"][5] == o5["argmax"][0] and same_bits(fa["best"][5:6], o5["best"])
    rng = np.random.default_rng(8)
    iou_ref = (ref + rng.normal(0, 12, ref.shape)).astype(np.float32)
    ex, fa, st = both(case, iou_boxes=T(iou_ref))
    check_fast(ex, fa)
    ex, fa, st = both((cubes, K, im, ref, mu, sg, None))      # no rectangle at all
    assert st[0] == 8
    check_fast(ex, fa)


def test_fast_is_the_default_of_the_argmax_only_launch():
    case = random_case(32, 1000, seed=77, nasty=False)
    cubes, K, im, ref, mu, sg, rect = case
    lean = geo.cubes_project_score(T(cubes), T(K), im, T(ref), T(mu), T(sg), T(rect), want=())
    o = og.project_and_score(*case)
    assert (lean["argmax"].cpu().numpy() == o["argmax"]).all()
    assert same_bits(lean["best"].cpu().numpy(), o["best"])


def test_fast_full_size_matches_exact_kernel():
    """BASELINE configs[2] size: 1024 objects x 1000 cubes, the bench's own inputs"""
    import bench
    inp = bench.geometry_inputs(1024, 1000, 1234, DEV)
    a = (inp["cubes"], inp["K"], inp["im_wh"], inp["ref"], inp["mu"], inp["sg"], inp["rect"])
    st = torch.zeros(2, dtype=torch.int64, device=DEV)
    ex = geo.cubes_project_score(*a, fast=False)
    fa = geo.cubes_project_score(*a, fast=True, stats=st)
    assert torch.equal(ex["argmax"], fa["argmax"])
    assert torch.equal(ex["best"].view(torch.int32), fa["best"].view(torch.int32))
    for k in PLANES:
        assert_close(fa[k].cpu().numpy(), ex[k].cpu().numpy(), rtol=1e-4, atol=2e-4 if k in ("corners", "boxes") else 1e-5)
    assert st[0].item() < 64, st


def test_g2_golden_fast_kernel(golden_dir):
    """the reference's own outputs (G2) against the fast kernel: planes at the tolerances the exact kernel is held to, argmax =="""
    g = load(golden_dir, "geometry_g2_project_score.npz")
    st = torch.zeros(2, dtype=torch.int64, device=DEV)
    out = geo.cubes_project_score(T(g["cubes"]), T(g["K"]), tuple(g["im_wh"]), T(g["ref_boxes"]), T(g["prior_mu"]),
                                  T(g["prior_sigma"]), T(g["rect_pts"]), fast=True, stats=st)
    out = {k: v.cpu().numpy() for k, v in out.items() if v is not None}
    frac = np.mean(np.abs(out["corners"] - g["corners2d"]) <= 1e-4 * np.abs(g["corners2d"]) + 1e-3)
    assert frac > 0.999
    assert_close(out["corners"], g["corners2d"], rtol=1e-4, atol=2e-2)
    assert_close(out["iou"], g["iou"], atol=2e-5)
    assert_close(out["dim"], g["dim"], atol=2e-5)
    assert_close(out["corner"], g["corner"], atol=2e-5)
    assert_close(out["combined"], g["combined"], atol=1e-5)
    assert (out["argmax"] == g["argmax"]).all()
    o = og.project_and_score(g["cubes"], g["K"], tuple(g["im_wh"]), g["ref_boxes"], g["prior_mu"], g["prior_sigma"], g["rect_pts"])
    assert (out["argmax"] == o["argmax"]).all() and same_bits(out["best"], o["best"])


def test_fast_sub_pixel_boxes_take_the_exact_sequence():
    """cubes that project to a fraction of a pixel inside the reference box: the IoU of such a box is only good to
    delta / width -- those objects run the exact sequence"""
    cubes, K, im, ref, mu, sg, rect = random_case(4, 1000, seed=31, nasty=False)
    cubes[1, :, 3:6] = 0.0004                                 # ~0.05 px at these depths
    cubes[1, :, 0] = np.linspace(-0.2, 0.2, 1000); cubes[1, :, 1] = 0.0; cubes[1, :, 2] = 6.0
    ref[1] = [200.0, 200.0, 320.0, 320.0]
    case = (cubes, K, im, ref, mu, sg, rect)
    ex, fa, st = both(case)
    assert st[0] >= 1
    check_fast(ex, fa)
    o = og.project_and_score(*case)
    assert (fa["argmax"] == o["argmax"]).all() and same_bits(fa["best"], o["best"])


def test_out_buffers_are_reused():
    case = random_case(3, 1000, seed=41, nasty=False)
    cubes, K, im, ref, mu, sg, rect = case
    a = (T(cubes), T(K), im, T(ref), T(mu), T(sg), T(rect))
    first = geo.cubes_project_score(*a, fast=True)
    ptrs = {k: v.data_ptr() for k, v in first.items() if v is not None}
    keep = {k: v.clone() for k, v in first.items() if v is not None}
    for v in first.values():
        if v is not None:
            v.zero_()
    again = geo.cubes_project_score(*a, fast=True, out=first)
    assert again is first and {k: v.data_ptr() for k, v in again.items() if v is not None} == ptrs
    for k, v in keep.items():
        assert torch.equal(again[k], v), k
    with pytest.raises(ValueError):
        geo.cubes_project_score(*a, fast=True, want=(), out=first)
