"""CPU: the numpy oracle (oracle/geometry.py) against the golden vectors that
were generated from the reference itself (tests/golden/make_golden_geometry.py).
Tolerances: 1e-4 relative for float32 geometry (north_star), argmax exact."""
import os

import numpy as np
import pytest

from oracle import geometry as og

RTOL = 1e-4


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def assert_close(a, b, rtol=RTOL, atol=1e-5):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert (nan_a == nan_b).all(), "NaN pattern differs"
    inf_a, inf_b = np.isinf(a), np.isinf(b)
    assert (inf_a == inf_b).all() and (a[inf_a] == b[inf_b]).all()
    m = ~(nan_a | inf_a)
    np.testing.assert_allclose(a[m], b[m], rtol=rtol, atol=atol)


def test_exp_f32_matches_libm():
    x = np.concatenate([-np.logspace(-6, 1.9, 4000), [0.0, -87.0, -86.9, -100.0, -np.inf]]).astype(np.float32)
    ours = og.exp_f32(x)
    ref = np.exp(x.astype(np.float64))
    ok = x >= -87
    rel = np.abs(ours[ok] - ref[ok]) / ref[ok]
    assert rel.max() < 3e-7
    assert (ours[~ok] == 0).all()
    assert np.isnan(og.exp_f32(np.array([np.nan], np.float32)))[0]


def test_g1_corners(golden_dir):
    g = load(golden_dir, "geometry_g1_corners.npz")
    v = og.cuboid_corners(g["box6"], g["R"])
    assert_close(v, g["verts"], atol=1e-5)


def test_g2_corners_projection_boxes(golden_dir):
    g = load(golden_dir, "geometry_g2_project_score.npz")
    c3 = og.cubes_corners(g["cubes"])
    assert_close(c3, g["corners3d"], atol=2e-5)
    c2 = og.project_corners(c3, g["K"], tuple(g["im_wh"]))
    # projection of near-camera cubes amplifies rounding; those saturate at the clamp
    assert_close(c2, g["corners2d"], rtol=RTOL, atol=2e-2)
    frac_exact = np.mean(np.abs(c2 - g["corners2d"]) <= 1e-4 * np.abs(g["corners2d"]) + 1e-3)
    assert frac_exact > 0.999
    bx = og.corners_to_boxes(g["corners2d"])
    assert (bx == g["boxes"]).all()


def test_g3_scores_and_argmax(golden_dir):
    g = load(golden_dir, "geometry_g2_project_score.npz")
    N, P = g["cubes"].shape[:2]
    for i in range(N):
        iou = og.iou_one_to_many(g["ref_boxes"][i], g["boxes"][i])
        assert_close(iou, g["iou"][i], atol=1e-6)
        dim, _, _, _ = og.score_dimensions(g["prior_mu"][i], g["prior_sigma"][i], g["cubes"][i, :, 3:6],
                                           g["ref_boxes"][i], g["boxes"][i])
        assert_close(dim, g["dim"][i], atol=1e-6)
        s = og.corner_chamfer(g["rect_pts"][i], g["corners2d"][i])
        assert_close(s, g["chamfer"][i], rtol=1e-6, atol=1e-4)
        cor, _, _ = og.score_corners_from_rect(g["rect_pts"][i], g["corners2d"][i])
        assert_close(cor, g["corner"][i], atol=1e-6)
    out = og.project_and_score(g["cubes"], g["K"], tuple(g["im_wh"]), g["ref_boxes"], g["prior_mu"],
                               g["prior_sigma"], g["rect_pts"])
    assert_close(out["combined"], g["combined"], atol=1e-6)
    assert (out["argmax"] == g["argmax"]).all()


def test_chamfer_against_scipy():
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(0)
    rect = rng.uniform(0, 512, (4, 2)).astype(np.float32)
    c2 = rng.uniform(-200, 900, (200, 8, 2)).astype(np.float32)
    ours = og.corner_chamfer(rect, c2)
    ref = np.array([np.mean(cKDTree(c2[j]).query(rect)[0]) for j in range(200)]).astype(np.float32)
    assert (ours == ref).all()


def test_iou_hand_cases():
    # detectron2 pairwise_iou restated; hand-computed known answers (parity unpinned vs detectron2)
    r = np.array([0, 0, 10, 10], np.float32)
    b = np.array([[0, 0, 10, 10], [5, 5, 15, 15], [10, 10, 20, 20], [20, 20, 30, 30], [2, 2, 4, 4]], np.float32)
    iou = og.iou_one_to_many(r, b)
    np.testing.assert_allclose(iou, [1.0, 25 / 175, 0.0, 0.0, 4 / 100], rtol=1e-6)


def test_g5_yaw_table(golden_dir):
    g = load(golden_dir, "geometry_g5_yaw_table.npz")
    for n, t in zip(g["normals"], g["tables"]):
        assert_close(og.yaw_table(n), t, atol=2e-6)


def test_g6_propose(golden_dir):
    g = load(golden_dir, "geometry_g6_propose.npz")
    normals = g["normals"]            # (D,N,P) in draw order
    N = g["boxes"].shape[0]; P = int(g["P"])
    cubes_ref = g["cubes"]
    # split the recorded draws: the last three are x,y,z; the rest are w/h/l rounds.
    ctr = normals[-3:]
    dims_draws = normals[:-3]
    mu, sg = g["prior_mu"], g["prior_sigma"]
    # replay the rejection loop to find how many rounds each of w,h,l consumed
    def rounds(mean, std, lo, hi, draws):
        s = mean[:, None] + std[:, None] * draws[0]; r = 1
        while ((s < lo) | (s > hi[:, None])).any():
            bad = (s < lo) | (s > hi[:, None])
            s = np.where(bad, mean[:, None] + std[:, None] * draws[r], s); r += 1
        return r
    f = np.float32
    rw = rounds(mu[:, 0], sg[:, 0], f(0.05), mu[:, 0] + f(2) * sg[:, 0], dims_draws)
    rh = rounds(mu[:, 1], sg[:, 1] * f(1.1), f(0.05), mu[:, 1] + f(2.2) * sg[:, 1], dims_draws[rw:])
    rl = rounds(mu[:, 2], sg[:, 2], f(0.05), mu[:, 2] + f(2) * sg[:, 2], dims_draws[rw + rh:])
    assert rw + rh + rl == dims_draws.shape[0]
    R = max(rw, rh, rl)
    dn = np.zeros((R, 3, N, P), np.float32)
    dn[:rw, 0] = dims_draws[:rw]; dn[:rh, 1] = dims_draws[rw:rw + rh]; dn[:rl, 2] = dims_draws[rw + rh:]
    cubes = og.propose_from_draws(g["boxes"], g["depth"], mu, sg, g["K"], P, dn, ctr, g["yaw_idx"], g["normal"])
    assert_close(cubes[..., 3:], cubes_ref[..., 3:], atol=2e-6)       # dims + rotation
    assert_close(cubes[..., :3], cubes_ref[..., :3], rtol=2e-4, atol=2e-4)   # centres (median/std reductions)


def test_g8_ransac(golden_dir):
    g = load(golden_dir, "geometry_g8_ransac.npz")
    neg_eq, cnt, best, counts = og.ransac_plane(g["pts"], g["triples"], float(g["thresh"]))
    assert cnt == int(g["n_inliers"])
    assert_close(neg_eq, g["neg_equation"], atol=1e-6)


def test_argmax_nan_first():
    x = np.array([0.1, np.nan, 0.9, np.nan], np.float32)
    assert og.argmax_numpy(x) == 1
    assert og.argmax_numpy(np.array([0.3, 0.9, 0.9], np.float32)) == 1


def test_empty_objects():
    out = og.project_and_score(np.zeros((0, 1000, 15), np.float32), np.eye(3, dtype=np.float32), (512, 512),
                               np.zeros((0, 4), np.float32), np.zeros((0, 3), np.float32),
                               np.zeros((0, 3), np.float32), np.zeros((0, 4, 2), np.float32))
    assert out["argmax"].shape == (0,)
