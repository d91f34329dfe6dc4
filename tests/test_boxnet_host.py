"""CPU: host / tensor logic of the batched BoxNet path that needs no kernel -- the sync-free RANSAC triple sampler, the
batched back-projection, the ground-normal fix-ups, the rejection-round hint."""
import importlib

import torch

plane = importlib.import_module("3dod_amd.ProposalNetwork.utils.plane")
boxer = importlib.import_module("3dod_amd.cubercnn.modeling.roi_heads.boxer")
PN = importlib.import_module("3dod_amd.ProposalNetwork.proposals.proposals")


def test_sample_triples_batched_distinct_and_eligible():
    g = torch.Generator().manual_seed(0)
    B, Q, T = 4, 500, 2000
    elig = torch.zeros(B, Q, dtype=torch.bool)
    elig[0, 100:400] = True
    elig[1, ::7] = True
    elig[2, [3, 250, 499]] = True                      # exactly three eligible points: every triple is a permutation of them
    elig[3] = True
    tri = plane.Plane.sample_triples_batched(elig, B, Q, T, "cpu", g).long()
    assert tri.shape == (B, T, 3) and tri.min() >= 0 and tri.max() < Q
    assert torch.gather(elig, 1, tri.reshape(B, -1)).all()
    assert ((tri[..., 0] != tri[..., 1]) & (tri[..., 0] != tri[..., 2]) & (tri[..., 1] != tri[..., 2])).all()
    assert set(tri[2].reshape(-1).tolist()) == {3, 250, 499}
    # every eligible point of image 0 is drawn about T*3/300 = 20 times
    hist = torch.bincount(tri[0].reshape(-1), minlength=Q)
    assert hist[:100].sum() == 0 and hist[400:].sum() == 0 and hist[100:400].min() >= 3 and hist[100:400].max() <= 50
    # eligible=None: all n_points points
    tri = plane.Plane.sample_triples_batched(None, 2, 50, 300, "cpu", g).long()
    assert tri.shape == (2, 300, 3) and tri.max() < 50 and len(set(tri.reshape(-1).tolist())) == 50


def test_depth_to_points_batched_equals_per_image():
    g = torch.Generator().manual_seed(1)
    depth = torch.rand(3, 40, 60, generator=g) * 5 + 0.5
    K = torch.tensor([[[50., 0, 30], [0, 55, 20], [0, 0, 1]], [[70., 0, 28], [0, 70, 22], [0, 0, 1]],
                      [[45., 0, 31], [0, 47, 19], [0, 0, 1]]])
    pts = boxer.depth_to_points(depth, K)
    assert pts.shape == (3, 8, 12, 3)
    for i in range(3):
        assert torch.equal(pts[i], boxer.depth_to_points(depth[i], K[i]))
    # the reference's formula (roi_heads.py:345-356): strided pixel indices with the full-resolution intrinsics
    v, u = 3, 7
    z = depth[1, 5 * v, 5 * u]
    assert torch.allclose(pts[1, v, u], torch.stack(((u - 28.0) * z / 70.0, (v - 22.0) * z / 70.0, z)))


def test_fix_ground_normal_batched_equals_per_vector():
    g = torch.Generator().manual_seed(2)
    n = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=1)
    both = boxer.fix_ground_normal(n.t()).t()
    for i in range(64):
        one = boxer.fix_ground_normal(n[i])
        assert torch.equal(both[i], one)
        assert one[1] >= 0 and abs(float(one.norm()) - 1) < 1e-6
        assert one[1].abs() >= one[0].abs() - 1e-6      # after the fix-ups the up axis dominates the side axis


def test_rejection_round_hint_doubles_and_is_bounded():
    old = PN._rounds_hint[0]
    try:
        PN._rounds_hint[0] = PN.ROUNDS
        PN.note_exhausted()
        assert PN._rounds_hint[0] == 2 * PN.ROUNDS
        PN._rounds_hint[0] = 8192
        try:
            PN.note_exhausted()
            raise AssertionError("expected the sampler to give up")
        except RuntimeError:
            pass
    finally:
        PN._rounds_hint[0] = old
