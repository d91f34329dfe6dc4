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
