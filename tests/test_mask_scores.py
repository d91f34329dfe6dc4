"""MABO mask scores: the oracle's raster definition on hand-computable polygons (CPU); the kernel against the
oracle is in tests/test_gpu_proposalnetwork.py.  cv2 (convexHull / fillPoly) is absent: parity with it is unpinned."""
import numpy as np

from oracle import geometry as og


def rect_points(x0, y0, x1, y1):
    # 4 corners + 4 points inside: the hull is the rectangle
    return np.array([[x0, y0], [x1, y0], [x1, y1], [x0, y1], [(x0 + x1) / 2, (y0 + y1) / 2], [x0 + 1, y0 + 1],
                     [x1 - 1, y1 - 1], [(x0 + x1) / 2, y0 + 1]], dtype=np.float32)


def test_rectangle_counts():
    mask = np.zeros((64, 96), dtype=bool)
    mask[:, :40] = True
    pts = np.stack([rect_points(10.7, 6.2, 50.9, 30.5),      # truncates to [10, 50] x [6, 30]
                    rect_points(0.0, 0.0, 95.0, 63.0),       # the whole canvas
                    rect_points(41.0, 1.0, 43.9, 3.9)])      # no grid sample inside: x in {41..43}, y in {1..3}
    c = og.segment_counts(pts, mask, 4)
    # samples x in {12,...,48} (10), y in {8,...,28} (6); mask covers x <= 39 -> x in {12,...,36} (7)
    assert c[0].tolist() == [60, 42]
    assert c[1].tolist() == [24 * 16, 10 * 16]
    assert c[2].tolist() == [0, 0]


def test_triangle_and_degenerate():
    mask = np.ones((40, 40), dtype=bool)
    tri = np.array([[0, 0], [32, 0], [0, 32], [1, 1], [2, 2], [8, 8], [4, 4], [10, 3]], dtype=np.float32)
    c = og.segment_counts(tri[None], mask, 4)
    # samples (4i, 4j) with i + j <= 8, i, j >= 0 (closed triangle): 45
    assert c[0].tolist() == [45, 45]
    seg = np.array([[4, 8]] * 4 + [[20, 8]] * 4, dtype=np.float32)         # a segment: samples on it count
    assert og.segment_counts(seg[None], mask, 4)[0].tolist() == [5, 5]
    pt = np.array([[8, 12]] * 8, dtype=np.float32)                          # a single point on the grid
    assert og.segment_counts(pt[None], mask, 4)[0].tolist() == [1, 1]
    nan = tri.copy()
    nan[3, 0] = np.nan
    assert og.segment_counts(nan[None], mask, 4)[0].tolist() == [0, 0]
