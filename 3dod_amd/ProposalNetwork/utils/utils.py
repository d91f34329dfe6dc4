"""Samplers and rotation builders of ProposalNetwork/utils/utils.py (the pieces the proposal path uses)."""
import torch

from ...d2lite import pairwise_iou


class Draws:
    """Source of the random variates of the proposal samplers.  The reference calls torch.rand / torch.randn /
    torch.normal / torch.randperm inline; routing them through one object keeps the call ORDER and SHAPES of the reference
    (the recorded-draws goldens replay them, tests/golden/make_golden_proposals.py) and lets callers pass a generator."""

    def __init__(self, generator=None):
        self.generator = generator

    def _g(self, device):
        g = self.generator
        return g if (g is not None and torch.device(g.device).type == torch.device(device).type) else None

    def rand(self, shape, device):
        return torch.rand(tuple(shape), device=device, generator=self._g(device))

    def randn(self, shape, device="cpu"):
        return torch.randn(tuple(shape), device=device, generator=self._g(device))

    def normal(self, means, stds):
        return torch.normal(means, stds, generator=self._g(means.device))

    def randperm(self, n):
        return torch.randperm(n, generator=self._g("cpu"))


def sample_normal_in_range(means, stds, count, threshold_low=None, threshold_high=None, rng=None):
    """utils.py:42-60: (N,) means / stds -> (N,count) normal samples; with thresholds, samples outside [low, high] are
    redrawn (a full (N,count) draw per round, only the invalid entries replaced) for up to 10 000 rounds."""
    rng = rng or Draws()
    m, s = means.unsqueeze(1).expand(-1, count), stds.unsqueeze(1).expand(-1, count)
    samples = rng.normal(m, s)
    if threshold_high is not None and threshold_low is not None:
        hi = threshold_high.unsqueeze(1).expand_as(samples)
        tries = 0
        while True:
            invalid = (samples < threshold_low) | (samples > hi)
            if not bool(invalid.any()):
                break
            samples[invalid] = rng.normal(m, s)[invalid]
            tries += 1
            if tries == 10000:
                break
    return samples


def randn_orthobasis_torch(num_samples=1, num_instances=1, rng=None):
    """utils.py:62-69: (num_instances, num_samples, 3, 3) random rotations-up-to-sign: Gaussian rows normalised, row 0 =
    normalize(row1 x row2), row 1 = normalize(row2 x row0) (drawn on the host like the reference)"""
    rng = rng or Draws()
    z = rng.randn((num_instances, num_samples, 3, 3), "cpu")
    z = z / torch.norm(z, p=2, dim=-1, keepdim=True)
    z[:, :, 0] = torch.linalg.cross(z[:, :, 1], z[:, :, 2], dim=-1)
    z[:, :, 0] = z[:, :, 0] / torch.norm(z[:, :, 0], dim=-1, keepdim=True)
    z[:, :, 1] = torch.linalg.cross(z[:, :, 2], z[:, :, 0], dim=-1)
    z[:, :, 1] = z[:, :, 1] / torch.norm(z[:, :, 1], dim=-1, keepdim=True)
    return z


def gt_in_norm_range(range_, gt):
    """utils.py:149-153: position of gt inside [range_[0], range_[1]] in units of the range's length"""
    return (gt - range_[0]) / abs(range_[1] - range_[0])


def normalize_vector(v):
    v_mag = torch.sqrt(v.pow(2).sum())
    v_mag = torch.max(v_mag, torch.tensor([1e-8], device=v.device))
    return v / v_mag


def vec_perp_t(vec):
    """utils.py:112-118."""
    a, b, c = vec
    if a == 0:
        return torch.stack([torch.zeros_like(c), c, -b])
    return normalize_vector(torch.stack([b, -a, torch.zeros_like(a)]))


def rotate_vector_t(v, k, theta):
    """Rodrigues, utils.py:134-146."""
    cos_theta, sin_theta = torch.cos(theta), torch.sin(theta)
    v2 = v.view(-1, 1)
    term1 = v2 * cos_theta
    term2 = torch.linalg.cross(k, v).view(-1, 1) * sin_theta
    term3 = (k * (k @ v)).view(-1, 1) * (1 - cos_theta)
    return term1 + term2 + term3


def orthobasis_from_normal_t(normal, yaw_angles):
    """utils.py:120-132: (n_yaw,3,3), columns (x_theta, normal, normal x x_theta)."""
    n = len(yaw_angles)
    x = rotate_vector_t(vec_perp_t(normal), normal, yaw_angles)
    y = torch.linalg.cross(normal.view(-1, 1).expand_as(x), x, dim=0)
    return torch.cat([x.t(), normal.unsqueeze(0).repeat(n, 1), y.t()], dim=1).reshape(n, 3, 3).transpose(2, 1)


def vectorized_linspace(start_tensor, end_tensor, number_of_steps):
    """utils.py:170-177."""
    spacing = (end_tensor - start_tensor) / (number_of_steps - 1)
    lin = torch.arange(start=0, end=number_of_steps, dtype=start_tensor.dtype, device=start_tensor.device)
    lin = lin.repeat(start_tensor.size(0), 1)
    return lin * spacing[:, None] + start_tensor[:, None]


def iou_2d(gt_box, proposal_boxes):
    """utils.py:186-192."""
    return pairwise_iou(gt_box, proposal_boxes).flatten()


def iou_3d(gt_cube, proposal_cubes):
    """utils.py:194-210: exact IoU3D of one ground-truth cube against the proposal cubes of an object -> (P,).
    pytorch3d.ops.box3d_overlap is replaced by cr_box3d_overlap (3dod_amd/csrc/iou3d.hip)."""
    from ... import geometry as geo
    gt_corners = gt_cube.get_all_corners()[0]
    proposal_corners = proposal_cubes.get_all_corners()[0]
    vol, iou = geo.box3d_overlap(gt_corners, proposal_corners)
    return iou[0]
