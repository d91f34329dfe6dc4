"""TEST INFRASTRUCTURE (oracle): the per-object mask -> 4-point rectangle step of score_corners
(ProposalNetwork/scoring/scorefunction.py:58-68: cv2.findContours(RETR_EXTERNAL) -> largest contour -> cv2.minAreaRect
-> cv2.boxPoints), restated with numpy/scipy because OpenCV is not installed: largest 8-connected component -> convex
hull of its pixel centres -> minimum-area enclosing rectangle by rotating calipers.  The checker of cr_mask_rects.
[third-party: parity unpinned w.r.t. OpenCV -- cv2.contourArea ranks contours by polygon area, this ranks components by
pixel count; the corner order differs from cv2.boxPoints, which the order-invariant chamfer score does not see.]"""
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
    c, s = np.cos(ang)[:, None], np.sin(ang)[:, None]                      # every candidate edge direction at once
    rx = hull[None, :, 0] * c + hull[None, :, 1] * s
    ry = hull[None, :, 1] * c - hull[None, :, 0] * s
    x0, x1, y0, y1 = rx.min(1), rx.max(1), ry.min(1), ry.max(1)
    k = int(np.argmin((x1 - x0) * (y1 - y0)))                              # first minimum in angle order
    c, s = c[k, 0], s[k, 0]
    box = np.array([[x0[k], y0[k]], [x1[k], y0[k]], [x1[k], y1[k]], [x0[k], y1[k]]])
    return (box @ np.array([[c, s], [-s, c]])).astype(np.float32)


def rect_from_mask(mask: np.ndarray):
    """(H,W) bool/uint8 mask -> (4,2) float32 (x,y) rectangle of its largest component, or None if empty."""
    m = np.asarray(mask).astype(bool)
    if not m.any():
        return None
    rows, cols = np.nonzero(m.any(1))[0], np.nonzero(m.any(0))[0]
    r0, c0 = rows[0], cols[0]
    m = m[r0:rows[-1] + 1, c0:cols[-1] + 1]                                # label only the mask's bounding window
    lab, n = ndimage.label(m, structure=np.ones((3, 3)))
    if n > 1:
        sizes = ndimage.sum(m, lab, index=np.arange(1, n + 1))
        m = lab == (1 + int(np.argmax(sizes)))
    # the hull of a pixel set is the hull of each occupied row's first and last pixel: <= 2H points reach the sort
    # and Qhull instead of every mask pixel
    rows = np.nonzero(m.any(1))[0]
    sub = m[rows]
    x0 = sub.argmax(1)
    x1 = m.shape[1] - 1 - sub[:, ::-1].argmax(1)
    return min_area_rect(np.concatenate([np.stack([x0 + c0, rows + r0], 1), np.stack([x1 + c0, rows + r0], 1)]))
