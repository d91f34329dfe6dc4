"""CPU: the oracle of the mask -> minimum-area rectangle step (cv2.minAreaRect stand-in; parity unpinned w.r.t. OpenCV)."""
import numpy as np

from oracle import rect


def _area(r):
    a, b = np.linalg.norm(r[1] - r[0]), np.linalg.norm(r[2] - r[1])
    return a * b


def test_axis_aligned_and_largest_component():
    m = np.zeros((60, 80), bool)
    m[10:30, 20:60] = True            # 20 x 40 block -> pixel centres span 19 x 39
    m[50:53, 2:5] = True              # a smaller second component is ignored
    r = rect.rect_from_mask(m)
    assert sorted(np.round(r[:, 0]).tolist()) == [20, 20, 59, 59] and sorted(np.round(r[:, 1]).tolist()) == [10, 10, 29, 29]
    assert rect.rect_from_mask(np.zeros((5, 5), bool)) is None


def test_rotated_rectangle_is_tight():
    rng = np.random.default_rng(0)
    th = 0.5
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    corners = np.array([[-30, -10], [30, -10], [30, 10], [-30, 10]], float) @ R.T + [100, 80]
    pts = np.concatenate([corners, (rng.uniform(-1, 1, (300, 2)) * [30, 10]) @ R.T + [100, 80]])
    r = rect.min_area_rect(pts)
    assert abs(_area(r) - 60 * 20) < 1e-3
    # every input point lies inside the rectangle
    e0, e1 = r[1] - r[0], r[3] - r[0]
    u = (pts - r[0]) @ e0 / (e0 @ e0)
    v = (pts - r[0]) @ e1 / (e1 @ e1)
    assert (u > -1e-6).all() and (u < 1 + 1e-6).all() and (v > -1e-6).all() and (v < 1 + 1e-6).all()
