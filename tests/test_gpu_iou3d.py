"""GPU: cr_box3d_overlap (float32, one thread per pair) against the float64 oracle (oracle/iou3d.py), the reference's
known-answer fixture (0.9944) and the corners produced by the product's own get_cuboid_verts_faces kernel."""
import importlib

import numpy as np
import pytest
import torch

from oracle import iou3d as O
from tests.test_iou3d import REF_C1, REF_C2, REF_IOU, box, rot

pytestmark = pytest.mark.gpu
geo = importlib.import_module("3dod_amd.geometry")
DEV = torch.device("cuda:0")


def test_known_answer_and_analytic():
    vol, iou = geo.box3d_overlap(torch.tensor([REF_C1], device=DEV), torch.tensor([REF_C2], device=DEV))
    assert abs(float(iou[0, 0]) - REF_IOU) < 1e-3
    a = box([10, -3, 20], [2, 2, 2])
    bs = [box([11, -3, 20], [2, 2, 2]), box([11, -2, 21], [2, 2, 2]), box([10, -3, 20], [1, 1, 1]), box([13, -3, 20], [2, 2, 2]),
          box([12, -3, 20], [2, 2, 2]), a, box([10, -3, 20], [2, 2, 2], rot([0, 0, 1], np.pi / 4))]
    want = [4.0, 1.0, 1.0, 0.0, 0.0, 8.0, 16 * (np.sqrt(2) - 1)]
    vol, iou = geo.box3d_overlap(torch.tensor(a[None], dtype=torch.float32, device=DEV),
                                 torch.tensor(np.stack(bs), dtype=torch.float32, device=DEV))
    assert np.allclose(vol.cpu().numpy()[0], want, rtol=2e-5, atol=2e-4), vol
    assert abs(float(iou[0, 5]) - 1.0) < 1e-5


def test_random_pairs_vs_oracle():
    rng = np.random.default_rng(5)
    A = np.stack([box(rng.normal(size=3) + [0, 0, 8], rng.uniform(0.4, 3, 3), rot(rng.normal(size=3), rng.uniform(0, 3))) for _ in range(24)])
    B = np.stack([box(A[i % 24].mean(0) + rng.normal(size=3) * 0.6, rng.uniform(0.4, 3, 3), rot(rng.normal(size=3), rng.uniform(0, 3)))
                  for i in range(40)])
    vol, iou = geo.box3d_overlap(torch.tensor(A, dtype=torch.float32, device=DEV), torch.tensor(B, dtype=torch.float32, device=DEV))
    rvol, riou = O.box3d_overlap(A, B)
    assert np.abs(iou.cpu().numpy() - riou).max() < 2e-4, np.abs(iou.cpu().numpy() - riou).max()
    assert (riou > 0.05).sum() > 20 and (riou == 0).sum() > 20           # both regimes covered


def test_on_cuboid_corners_kernel():
    """corners from cr_cuboid_corners (math_util.get_cuboid_verts_faces order) feed box3d_overlap like iou_3d does."""
    g = torch.Generator().manual_seed(3)
    n = 16
    box6 = torch.cat([torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0.0, 0.0, 6.0]), torch.rand(n, 3, generator=g) + 0.5], 1)
    util = importlib.import_module("3dod_amd.cubercnn.util.math_util")
    R = util.rotation_6d_to_matrix(torch.randn(n, 6, generator=g))
    verts = geo.cuboid_corners(box6.to(DEV), R.to(DEV))
    vol, iou = geo.box3d_overlap(verts, verts)
    assert torch.allclose(torch.diagonal(iou).cpu(), torch.ones(n), atol=1e-4)
    assert torch.allclose(torch.diagonal(vol).cpu(), box6[:, 3:].prod(1), rtol=1e-4)
    rvol, riou = O.box3d_overlap(verts.cpu().numpy(), verts.cpu().numpy())
    assert np.abs(iou.cpu().numpy() - riou).max() < 2e-4


def test_ap3d_same_through_kernel_and_oracle():
    """SURVEY 8(d) "AP3D parity": the same detections scored with the HIP IoU3D kernel and with the float64 oracle give
    the same AP3D (1e-3)."""
    ev = importlib.import_module("3dod_amd.cubercnn.evaluation")
    rng = np.random.default_rng(11)
    gts, dts = [], []
    for img in range(6):
        for k in range(5):
            ctr = rng.normal(size=3) * [3, 1, 4] + [0, 0, 18]
            dims = rng.uniform(0.5, 3, 3)
            R = rot([0, 1, 0], rng.uniform(0, 3))
            cat = int(rng.integers(0, 3))
            c = box(ctr, dims, R)
            gts.append({"image_id": img, "category_id": cat, "id": len(gts) + 1, "bbox": [0, 0, 10, 10], "area": 100.0,
                        "bbox3D": c.tolist(), "depth": float(ctr[2]), "ignore2D": 0, "ignore3D": int(rng.random() < 0.1)})
            for j in range(2):                       # a good and a sloppy detection per object
                c2 = box(ctr + rng.normal(size=3) * (0.1 + 0.5 * j), dims * rng.uniform(0.8, 1.2, 3), R @ rot([0, 1, 0], rng.normal() * 0.2))
                dts.append({"image_id": img, "category_id": cat, "bbox": [0, 0, 10, 10], "area": 100.0, "bbox3D": c2.tolist(),
                            "depth": float(ctr[2]), "score": float(rng.random())})
    a = ev.Omni3Deval(gts, dts, "3D").evaluate().accumulate().summarize()              # cr_box3d_overlap
    b = ev.Omni3Deval(gts, dts, "3D", iou3d_fn=lambda d, g: O.box3d_overlap(np.asarray(d), np.asarray(g))[1]).evaluate().accumulate().summarize()
    assert 0.05 < b[0] < 0.95
    assert np.abs(a - b).max() < 1e-3, (a, b)
