"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy float64) of the exact IoU of two oriented 3D boxes given by their
8 corners, the quantity the reference takes from pytorch3d `box3d_overlap` / `_C.iou_box3d` [third-party, absent:
unpinned `@stable`, requirements.txt:8] at ProposalNetwork/utils/utils.py:194-210, cubercnn/evaluation/
omni3d_evaluation.py:155 and cubercnn/modeling/roi_heads/roi_heads.py:1563.

Algorithm (restated from the published definition of the problem, not from pytorch3d's source): the intersection of
two convex polyhedra is bounded by pieces of their faces.  Each of the 6 quad faces of box A is clipped (Sutherland-
Hodgman) against the 6 half-spaces of box B and vice versa; the volume follows from the divergence theorem,
V = 1/3 * sum over boundary polygons of (n . p0) * area.  Faces of A are clipped inclusively (by a tolerance); a face of
B is clipped exclusively against planes of A with the same orientation, so that coplanar faces of equal orientation are
counted exactly once while coplanar faces of opposite orientation (touching boxes) both stay and cancel.

Pinned by the reference's own known-answer test ProposalNetwork/utils/tests/test_iou.py:4-27 (IoU = 0.9944) in
tests/test_iou3d.py, plus analytic cases (axis-aligned overlaps, containment, disjoint, identical, rotated)."""
import numpy as np

# corner order of pytorch3d boxes (the order get_cuboid_verts_faces produces, math_util.py:198-207); the winding does not
# matter: normals are oriented with the box centre
FACES = np.array([[0, 1, 2, 3], [3, 2, 6, 7], [0, 1, 5, 4], [0, 3, 7, 4], [1, 2, 6, 5], [4, 5, 6, 7]])


def box_planes(c):
    """c (8,3) -> outward unit normals (6,3) and offsets (6,) with n.x <= off inside."""
    ctr = c.mean(0)
    a = c[FACES[:, 0]]
    n = np.cross(c[FACES[:, 1]] - a, c[FACES[:, 3]] - a)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    flip = (n * (ctr - a)).sum(1) > 0
    n[flip] *= -1
    return n, (n * a).sum(1)


def box_volume(c):
    return abs(np.linalg.det(np.stack([c[1] - c[0], c[3] - c[0], c[4] - c[0]])))


def _clip(poly, n, off, shift):
    """keep the part of the polygon with n.x - off <= shift."""
    out = []
    m = len(poly)
    if m == 0:
        return out
    d = [float(np.dot(n, p) - off - shift) for p in poly]
    for i in range(m):
        j = (i + 1) % m
        if d[i] <= 0:
            out.append(poly[i])
            if d[j] > 0:
                out.append(poly[i] + (poly[j] - poly[i]) * (d[i] / (d[i] - d[j])))
        elif d[j] <= 0:
            out.append(poly[i] + (poly[j] - poly[i]) * (d[i] / (d[i] - d[j])))
    return out


def _face_term(poly, n):
    """(n . p0) * area of a planar polygon with unit normal n."""
    if len(poly) < 3:
        return 0.0
    p0 = poly[0]
    s = np.zeros(3)
    for i in range(1, len(poly) - 1):
        s += np.cross(poly[i] - p0, poly[i + 1] - p0)
    return float(np.dot(n, p0)) * 0.5 * abs(float(np.dot(s, n)))


def intersection_volume(c1, c2, tol=None):
    c1 = np.asarray(c1, np.float64)
    c2 = np.asarray(c2, np.float64)
    o = c1.mean(0)                                     # work relative to box 1's centre
    c1, c2 = c1 - o, c2 - o
    if tol is None:
        # corners usually come from float32 arithmetic (not exactly planar quads): the coplanarity tolerance has to
        # cover that noise, or nearly identical boxes lose faces; the volume error it introduces is O(tol * area)
        tol = 2e-6 * max(np.abs(c1).max(), np.abs(c2).max(), 1e-30)
    n1, o1 = box_planes(c1)
    n2, o2 = box_planes(c2)
    vol = 0.0
    for f in range(6):                                  # faces of box 1 inside box 2 (inclusive)
        poly = [c1[i] for i in FACES[f]]
        for k in range(6):
            poly = _clip(poly, n2[k], o2[k], +tol)
        vol += _face_term(poly, n1[f])
    for f in range(6):                                  # faces of box 2 inside box 1; a face coplanar with a face of box 1
        poly = [c2[i] for i in FACES[f]]                # of the SAME orientation was already counted there -> exclusive;
        for k in range(6):                              # opposite orientation (touching boxes) must stay: the two cancel
            same = float(np.dot(n2[f], n1[k])) > 0.999
            poly = _clip(poly, n1[k], o1[k], -tol if same else +tol)
        vol += _face_term(poly, n2[f])
    return max(vol / 3.0, 0.0)


def box3d_overlap(b1, b2):
    """b1 (N,8,3), b2 (M,8,3) -> vol (N,M), iou (N,M) like pytorch3d.ops.box3d_overlap."""
    b1, b2 = np.asarray(b1, np.float64), np.asarray(b2, np.float64)
    vol = np.zeros((len(b1), len(b2)))
    iou = np.zeros_like(vol)
    v1 = [box_volume(c) for c in b1]
    v2 = [box_volume(c) for c in b2]
    for i, c1 in enumerate(b1):
        for j, c2 in enumerate(b2):
            v = intersection_volume(c1, c2)
            vol[i, j] = v
            iou[i, j] = v / (v1[i] + v2[j] - v)
    return vol, iou
