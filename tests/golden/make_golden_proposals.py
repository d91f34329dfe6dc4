"""Golden vectors for the ablation samplers of the proposal method: the REFERENCE'S OWN propose_random / propose_xy_patch /
propose_z / propose_random_dim / propose_aspect_ratio / propose_random_rotation, `propose` without a ground normal,
`statistics` and randn_orthobasis_torch (ProposalNetwork/proposals/proposals.py:20-336,427-447; utils/utils.py:42-69),
run in the build container under the stub finder (_refimport.py).  Every random draw the reference makes (torch.rand /
randn / normal / randperm) is RECORDED in call order; the fixture holds the draws and the resulting cubes, the test replays
the draws through this repo's samplers (utils.Draws interface) and must reproduce the cubes.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_proposals.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _refimport  # noqa: E402

ref = _refimport.load()
P = ref.proposals
torch.set_num_threads(1)


class Recorder:
    NAMES = ("rand", "randn", "normal", "randperm")

    def __init__(self):
        self.log = []
        self.orig = {n: getattr(torch, n) for n in self.NAMES}

    def __enter__(self):
        for n in self.NAMES:
            def f(*a, _n=n, **k):
                out = self.orig[_n](*a, **k)
                self.log.append((_n, out.detach().cpu().clone()))
                return out
            setattr(torch, n, f)
        return self

    def __exit__(self, *exc):
        for n in self.NAMES:
            setattr(torch, n, self.orig[n])


def main():
    torch.manual_seed(11)
    N, Pn = 3, 48
    g = torch.Generator().manual_seed(5)
    ctr = torch.rand(N, 2, generator=g) * 100 + 80
    wh = torch.rand(N, 2, generator=g) * 60 + 30
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    depth = torch.rand(256, 256, generator=g) * 3 + 1.5
    K = torch.tensor([[260.0, 0, 128], [0, 260, 128], [0, 0, 1]])
    mu = torch.rand(N, 3, generator=g) * 0.8 + 0.4
    sg = 0.25 * mu
    # ground-truth cubes for `statistics`
    gt = torch.cat([torch.randn(N, 1, 3, generator=g), torch.rand(N, 1, 3, generator=g) + 0.3,
                    ref.utils.randn_orthobasis_torch(1, N).reshape(N, 1, 9)], 2)
    out = dict(boxes=boxes, depth=depth, K=K, prior_mu=mu, prior_sigma=sg, gt_cubes=gt, P=torch.tensor(Pn))
    rb = ref.Boxes(boxes)
    gtc = ref.spaces.Cubes(gt)
    cases = {"random": lambda: P.propose_random(rb, None, None, None, None, number_of_proposals=Pn, gt_cubes=gtc),
             "xy": lambda: P.propose_xy_patch(rb, None, None, (256, 256), K, number_of_proposals=Pn),
             "z": lambda: P.propose_z(rb, depth, None, (256, 256), None, number_of_proposals=Pn),
             "dim": lambda: P.propose_random_dim(rb, depth, (mu, sg), None, K, number_of_proposals=Pn),
             "aspect": lambda: P.propose_aspect_ratio(rb, depth, (mu, sg), None, K, number_of_proposals=Pn),
             "rotation": lambda: P.propose_random_rotation(rb, depth, (mu, sg), None, K, number_of_proposals=Pn, gt_cubes=gtc),
             "propose_no_normal": lambda: P.propose(rb, depth, (mu, sg), (256, 256), K, number_of_proposals=Pn, ground_normal=None)}
    for name, fn in cases.items():
        with Recorder() as r:
            cubes, stats, ranges = fn()
        out[f"{name}_cubes"] = cubes.tensor
        if stats is not None:
            out[f"{name}_stats"] = stats
            out[f"{name}_ranges"] = torch.as_tensor(np.asarray(ranges), dtype=torch.float32)
        out[f"{name}_ndraws"] = torch.tensor(len(r.log))
        for i, (kind, t) in enumerate(r.log):
            out[f"{name}_draw{i:03d}_{kind}"] = t
        print(name, "draws", len(r.log), "cubes", tuple(cubes.tensor.shape), "finite", bool(torch.isfinite(cubes.tensor).all()))
    npz = {k: v.detach().cpu().numpy() for k, v in out.items()}
    npz["notes"] = np.array("reference code run: ProposalNetwork/proposals/proposals.py:20-447, utils/utils.py:42-69, "
                            "utils/conversions.py:50-67, cubercnn/util/math_util.py:71-81 (mat2euler); stand-in: detectron2 Boxes")
    path = os.path.join(HERE, "proposals_variants.npz")
    np.savez_compressed(path, **npz)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
