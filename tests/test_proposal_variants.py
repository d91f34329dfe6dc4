"""The ablation samplers of the proposal method (ProposalNetwork/proposals/proposals.py:20-336, `propose` without a ground
normal :398-400, `statistics` :427-447, utils.randn_orthobasis_torch / sample_normal_in_range) against
tests/golden/proposals_variants.npz = the reference's own functions with every random draw recorded
(tests/golden/make_golden_proposals.py): replaying the draws through this repo's samplers must reproduce the cubes."""
import importlib
import os

import numpy as np
import pytest
import torch

PN = importlib.import_module("3dod_amd.ProposalNetwork.proposals.proposals")
U = importlib.import_module("3dod_amd.ProposalNetwork.utils.utils")
spaces = importlib.import_module("3dod_amd.ProposalNetwork.utils.spaces")
d2 = importlib.import_module("3dod_amd.d2lite")


class Replay:
    """utils.Draws interface fed from the recorded draws, in the reference's call order"""

    def __init__(self, g, name):
        n = int(g[f"{name}_ndraws"])
        keys = sorted(k for k in g.files if k.startswith(f"{name}_draw"))
        assert len(keys) == n
        self.items = [(k.rsplit("_", 1)[1], torch.from_numpy(g[k])) for k in keys]
        self.pos = 0

    def _next(self, kind, shape=None):
        k, t = self.items[self.pos]
        self.pos += 1
        assert k == kind, (self.pos, k, kind)
        if shape is not None:
            assert tuple(t.shape) == tuple(shape), (kind, t.shape, shape)
        return t.clone()

    def rand(self, shape, device):
        return self._next("rand", shape).to(device)

    def randn(self, shape, device="cpu"):
        return self._next("randn", shape).to(device)

    def normal(self, means, stds):
        return self._next("normal", means.shape).to(means.device)

    def randperm(self, n):
        return self._next("randperm", (n,))


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "proposals_variants.npz"), allow_pickle=False)


@pytest.mark.parametrize("name", ["random", "xy", "z", "dim", "aspect", "rotation", "propose_no_normal"])
def test_variant_reproduces_reference_given_its_draws(G, name):
    T = lambda k: torch.from_numpy(G[k])
    boxes, depth, K = d2.Boxes(T("boxes")), T("depth"), T("K")
    pri = (T("prior_mu"), T("prior_sigma"))
    P = int(G["P"])
    rng = Replay(G, name)
    gt = spaces.Cubes(T("gt_cubes")) if f"{name}_stats" in G.files else None
    if name == "propose_no_normal":
        cubes, stats, ranges = PN.propose_random_rotation(boxes, depth, pri, (256, 256), K, P, rng=rng)
    else:
        cubes, stats, ranges = PN.PROPOSAL_FUNCTIONS[name](boxes, depth, pri, (256, 256), K, number_of_proposals=P,
                                                           gt_cubes=gt, rng=rng)
    assert rng.pos == len(rng.items), "same number of random draws as the reference"
    np.testing.assert_allclose(cubes.tensor.numpy(), G[f"{name}_cubes"], rtol=1e-5, atol=1e-6)
    if gt is not None:
        np.testing.assert_allclose(stats.numpy(), G[f"{name}_stats"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(np.asarray(ranges, dtype=np.float32), G[f"{name}_ranges"], rtol=1e-5, atol=1e-6)


def test_propose_without_ground_normal_takes_the_random_basis_path():
    """`propose(ground_normal=None)` (proposals.py:398-400) no longer raises: it is propose_random_rotation"""
    g = torch.Generator().manual_seed(0)
    boxes = d2.Boxes(torch.tensor([[40.0, 50, 120, 140]]))
    depth = torch.rand(256, 256, generator=g) + 2
    pri = (torch.tensor([[0.6, 0.8, 0.7]]), torch.tensor([[0.1, 0.1, 0.1]]))
    K = torch.tensor([[260.0, 0, 128], [0, 260, 128], [0, 0, 1]])
    cubes, stats, ranges = PN.propose(boxes, depth, pri, (256, 256), K, number_of_proposals=32, ground_normal=None)
    R = cubes.tensor[0, :, 6:].reshape(-1, 3, 3)
    assert cubes.tensor.shape == (1, 32, 15) and stats is None
    # rows 0 and 1 are re-orthogonalised against row 2 (utils.py:62-69): unit rows, row0 ⟂ row1, row0 ⟂ row2
    assert torch.allclose(R.norm(dim=-1), torch.ones(32, 3), atol=1e-5)
    assert float((R[:, 0] * R[:, 1]).sum(-1).abs().max()) < 1e-5 and float((R[:, 0] * R[:, 2]).sum(-1).abs().max()) < 1e-5


def test_randn_orthobasis_matches_reference_formula():
    z = torch.randn(2, 5, 3, 3, generator=torch.Generator().manual_seed(3))

    class One:
        def randn(self, shape, device="cpu"):
            return z.clone()
    R = U.randn_orthobasis_torch(5, 2, One())
    zn = z / z.norm(dim=-1, keepdim=True)
    r0 = torch.linalg.cross(zn[:, :, 1], zn[:, :, 2]); r0 = r0 / r0.norm(dim=-1, keepdim=True)
    r1 = torch.linalg.cross(zn[:, :, 2], r0); r1 = r1 / r1.norm(dim=-1, keepdim=True)
    assert torch.allclose(R[:, :, 0], r0) and torch.allclose(R[:, :, 1], r1) and torch.allclose(R[:, :, 2], zn[:, :, 2])
