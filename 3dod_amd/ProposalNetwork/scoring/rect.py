"""The per-object mask -> 4-point rectangle step of score_corners (ProposalNetwork/scoring/scorefunction.py:58-68:
cv2.findContours(RETR_EXTERNAL) -> largest contour -> cv2.minAreaRect -> cv2.boxPoints), restated on the host with
numpy/scipy because OpenCV is not installed: largest 8-connected component -> convex hull of its pixel centres ->
minimum-area enclosing rectangle by rotating calipers.  [third-party: parity unpinned w.r.t. OpenCV; the corner
order differs from cv2.boxPoints, which the order-invariant chamfer score does not see.]"""
import numpy as np
from scipy import ndimage
from scipy.spatial import ConvexHull, QhullError


def min_area_rect(points: np.ndarray) -> np.ndarray:
    """(n,2) points -> (4,2) corners of the minimum-area enclosing rectangle."""
    pts = np.unique(np.asarray(points, dtype=np.float64), axis=0)
    if len(pts) == 1:
        return np.repeat(pts, 4, axis=0).astype(np.float32)
    try:
        hull = pts[ConvexHull(pts).vertices] if len(pts) >= 3 else pts
    except QhullError:                      # collinear
        hull = pts[[np.argmin(pts @ (pts[-1] - pts[0])), np.argmax(pts @ (pts[-1] - pts[0]))]]
    edges = np.roll(hull, -1, axis=0) - hull
    ang = np.unique(np.mod(np.arctan2(edges[:, 1], edges[:, 0]), np.pi / 2))
    best = None
    for a in ang:
        c, s = np.cos(a), np.sin(a)
        Rm = np.array([[c, s], [-s, c]])
        r = hull @ Rm.T
        mn, mx = r.min(0), r.max(0)
        area = (mx[0] - mn[0]) * (mx[1] - mn[1])
        if best is None or area < best[0]:
            box = np.array([[mn[0], mn[1]], [mx[0], mn[1]], [mx[0], mx[1]], [mn[0], mx[1]]]) @ Rm
            best = (area, box)
    return best[1].astype(np.float32)


def rect_from_mask(mask: np.ndarray):
    """(H,W) bool/uint8 mask -> (4,2) float32 (x,y) rectangle of its largest component, or None if empty."""
    m = np.asarray(mask).astype(bool)
    if not m.any():
        return None
    lab, n = ndimage.label(m, structure=np.ones((3, 3)))
    if n > 1:
        sizes = ndimage.sum(m, lab, index=np.arange(1, n + 1))
        m = lab == (1 + int(np.argmax(sizes)))
    ys, xs = np.nonzero(m)
    return min_area_rect(np.stack([xs, ys], 1))
