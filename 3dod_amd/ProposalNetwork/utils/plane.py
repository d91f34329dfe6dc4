"""Plane.fit_parallel -- ProposalNetwork/utils/plane.py:79-134 on cr_ransac_plane."""
import torch

from ... import geometry as geo


class Plane:
    def __init__(self):
        self.inliers = []
        self.equation = []

    @staticmethod
    def sample_triples(n_points, n_iter, device, generator=None):
        """three DISTINCT indices per iteration (random.sample(range(n), 3) in the reference)."""
        i0 = torch.randint(n_points, (n_iter,), device=device, generator=generator)
        i1 = (i0 + 1 + torch.randint(n_points - 1, (n_iter,), device=device, generator=generator)) % n_points
        i2 = torch.randint(n_points - 2, (n_iter,), device=device, generator=generator)
        lo, hi = torch.minimum(i0, i1), torch.maximum(i0, i1)
        i2 = i2 + (i2 >= lo).long()
        i2 = i2 + (i2 >= hi).long()
        return torch.stack((i0, i1, i2), 1).to(torch.int32)

    @staticmethod
    def sample_triples_batched(eligible, B, n_points, n_iter, device, generator=None):
        """sample_triples for B point sets at once without reading a count back: eligible (B,Q) bool or None (all of
        the n_points points); every set must have >= 3 eligible points.  Returns (B,n_iter,3) int32 indices into Q."""
        if eligible is None:
            cnt = torch.full((B, 1), n_points, device=device, dtype=torch.int64)
        else:
            cnt = eligible.sum(1, keepdim=True)
        u = torch.rand((3, B, n_iter), device=device, generator=generator, dtype=torch.float64)
        i0 = torch.minimum((u[0] * cnt).long(), cnt - 1)
        i1 = (i0 + 1 + torch.minimum((u[1] * (cnt - 1)).long(), cnt - 2)) % cnt
        i2 = torch.minimum((u[2] * (cnt - 2)).long(), cnt - 3)
        lo, hi = torch.minimum(i0, i1), torch.maximum(i0, i1)
        i2 = i2 + (i2 >= lo).long()
        i2 = i2 + (i2 >= hi).long()
        tri = torch.stack((i0, i1, i2), -1)                                  # ranks among the eligible points
        if eligible is not None:
            order = torch.sort(eligible.to(torch.uint8), dim=1, descending=True, stable=True).indices
            tri = torch.gather(order, 1, tri.reshape(B, -1)).reshape(B, n_iter, 3)
        return tri.to(torch.int32)

    def fit_parallel(self, pts: torch.Tensor, thresh=0.05, minPoints=100, maxIteration=1000, id_samples=None,
                     generator=None, need_inliers=True):
        """returns (-equation (4,), inlier indices) like the reference; need_inliers=False skips the index list (its
        length is data dependent, i.e. a host sync) for callers that only want the plane."""
        n_points = pts.shape[0]
        if id_samples is None:
            id_samples = self.sample_triples(n_points, maxIteration, pts.device, generator)
        neg_eq, counts, best = geo.ransac_plane(pts.float().contiguous(), id_samples, thresh, validate=False)
        eq = -neg_eq
        self.equation = eq
        if not need_inliers:
            self.inliers = None
            return neg_eq, None
        dist = (eq[0] * pts[:, 0] + eq[1] * pts[:, 1] + eq[2] * pts[:, 2] + eq[3]) / torch.sqrt(eq[0] ** 2 + eq[1] ** 2 + eq[2] ** 2)
        self.inliers = torch.where(torch.abs(dist) <= thresh)[0]
        self.equation = eq
        return neg_eq, self.inliers
