"""propose -- ProposalNetwork/proposals/proposals.py:338-424 of the reference.  The arithmetic runs in
cr_propose; this wrapper draws the random variates on the device (torch.randn / torch.randint, like the
reference's torch.normal / torch.randint) and redraws while the truncated-normal rejection rounds are exhausted
(sample_normal_in_range loops up to 10 000 rounds, utils.py:42-60)."""
import math

import torch

from ... import geometry as geo
from ..utils.spaces import Cubes

MIN_PROP_S = 0.05
ROUNDS = 8
_rounds_hint = [ROUNDS]


def propose(reference_box, depth_image, priors, im_shape, K, number_of_proposals=1, gt_cubes=None,
            ground_normal: torch.Tensor = None, generator=None):
    """reference_box: Boxes (N,4); depth_image (H,W); priors = (mean (N,3), std (N,3)) in (w,h,l); K (3,3).
    Returns (Cubes (N,P,15), stats, ranges); stats / ranges are None without gt_cubes, like the reference."""
    if ground_normal is None:
        # proposals.py:398-400: without a ground normal the orientation is a random orthonormal basis -- the rest is
        # `propose` itself, i.e. exactly propose_random_rotation (host-driven tensor expressions, not the fused kernel)
        return propose_random_rotation(reference_box, depth_image, priors, im_shape, K, number_of_proposals, gt_cubes,
                                       rng=_U.Draws(generator))
    boxes = reference_box.tensor
    dev = boxes.device
    N, P = boxes.shape[0], int(number_of_proposals)
    if N == 0:
        return Cubes(torch.zeros((0, P, 15), device=dev)), None, None
    mu, sg = priors[0].to(dev).float(), priors[1].to(dev).float()
    ctr = torch.randn((3, N, P), device=dev, generator=generator)
    yaw = torch.randint(36, (N, P), device=dev, generator=generator, dtype=torch.int32)
    rounds = ROUNDS
    for attempt in range(12):
        dn = torch.randn((rounds, 3, N, P), device=dev, generator=generator)
        cubes, exhausted = geo.propose_from_draws(boxes.contiguous(), depth_image.to(dev).float().contiguous(), mu, sg,
                                                  K.to(dev).float().contiguous(), P, dn, ctr, yaw,
                                                  ground_normal.to(dev).float().contiguous())
        if int(exhausted.item()) == 0:
            break
        rounds *= 2
    else:
        raise RuntimeError("truncated-normal rejection sampling did not converge (prior std too large for its range?)")
    if gt_cubes is None:
        return Cubes(cubes), None, None
    # proposals.py:416-424: where the ground truth lies inside the sampled ranges, and the ranges the offset statistics of
    # the MABO branch are normalised by (spread of the sampled centres, prior spreads of the dimensions, pi for the angles)
    x, y, z, w, h, l = cubes[..., :6].unbind(2)
    pi = torch.full((N,), math.pi, device=dev)
    ranges = torch.stack([x.std(dim=1) * 1.2, y.std(dim=1) * 0.8, z.std(dim=1) * 1.2, sg[:, 0], sg[:, 1] * 1.1, sg[:, 2],
                          pi, pi, pi], dim=1).cpu().numpy()
    return Cubes(cubes), statistics(gt_cubes, x, y, z, w, h, l), ranges


def propose_batched(boxes, img_idx, depth_images, priors, K, number_of_proposals, ground_normals, generator=None,
                    defer_check=False):
    """propose() for the objects of a whole batch in one launch (cr_propose_batched): boxes (N,4) tensor, img_idx (N)
    int32 image of each object, depth_images (B,H,W), priors = (mean (N,3), std (N,3)), K (B,3,3), ground_normals (B,3).
    Returns the (N,P,15) cube tensor after checking the rejection sampler's exhausted flag (one host sync, redraw with
    twice the rounds if needed).  defer_check=True returns (cubes, exhausted) without reading the flag, for callers that
    enqueue the rest of their pipeline first and call `note_exhausted` + retry only when it turns out non-zero."""
    dev = boxes.device
    N, P = boxes.shape[0], int(number_of_proposals)
    if N == 0:
        z = torch.zeros((0, P, 15), device=dev)
        return (z, torch.zeros((1,), dtype=torch.int32, device=dev)) if defer_check else z
    mu, sg = priors[0].to(dev).float(), priors[1].to(dev).float()
    depth = depth_images.to(dev).float().contiguous()
    ctr = torch.randn((3, N, P), device=dev, generator=generator)
    yaw = torch.randint(36, (N, P), device=dev, generator=generator, dtype=torch.int32)
    for attempt in range(12):
        dn = torch.randn((_rounds_hint[0], 3, N, P), device=dev, generator=generator)
        cubes, exhausted = geo.propose_from_draws_batched(boxes.float().contiguous(), img_idx, depth, mu, sg,
                                                          K.to(dev).float().contiguous(), P, dn, ctr, yaw,
                                                          ground_normals.to(dev).float().contiguous())
        if defer_check:
            return cubes, exhausted
        if int(exhausted.item()) == 0:
            return cubes
        note_exhausted()
    raise RuntimeError("truncated-normal rejection sampling did not converge (prior std too large for its range?)")


def note_exhausted():
    """the last propose ran out of rejection rounds: wide priors reject often, so start the next one with twice as many"""
    if _rounds_hint[0] >= 8192:
        raise RuntimeError("truncated-normal rejection sampling did not converge (prior std too large for its range?)")
    _rounds_hint[0] *= 2


# ---------------------------------------------------------------------------------------------------------------
# The ablation samplers of the reference (proposals.py:20-336) and the coverage statistics (:427-447).  They are
# selected by name through ROIHeads_Boxer.predict_cubes(proposal_function=...) (roi_heads.py:283-302) and are not on the
# hot path: plain tensor expressions on the boxes' device, written in the reference's order of operations AND of random
# draws (utils.Draws), so that the recorded-draws goldens replay them (tests/golden/proposals_variants.npz).
# ---------------------------------------------------------------------------------------------------------------
from ..utils import utils as _U                      # noqa: E402
from ..utils.conversions import pixel_to_normalised_space    # noqa: E402


def rescale_interval(x, min, max):
    """proposals.py:12-14: (min - max) * x + max"""
    return (min - max) * x + max


def lin_fun(x, coef):
    """proposals.py:16-18"""
    return coef[0] * x + coef[1]


def _finish(xyzwhl, rot, gt_cubes, ranges=None):
    cubes = Cubes(torch.cat((xyzwhl, rot.flatten(start_dim=2).to(xyzwhl.device)), dim=2))
    if gt_cubes is None:
        return cubes, None, None
    x, y, z, w, h, l = xyzwhl.unbind(2)
    stats = statistics(gt_cubes, x, y, z, w, h, l)
    return cubes, stats, (torch.ones(cubes.num_instances, 9) if ranges is None else ranges)


def _rand_dims(rng, n, P, dev):
    return [rescale_interval(rng.rand((n, P), dev), MIN_PROP_S, 2) for _ in range(3)]


def propose_random(reference_box, depth_image, priors, im_shape, K, number_of_proposals=1, gt_cubes=None,
                   ground_normal=None, rng=None):
    """proposals.py:20-45: everything uniform: x in [-2,2], y in [-1,1], z in [1,5], dims in [0.05,2], random basis"""
    rng = rng or _U.Draws()
    n, P, dev = len(reference_box), number_of_proposals, reference_box.device
    x = rng.rand((n, P), dev) * 4 - 2
    y = rng.rand((n, P), dev) * 2 - 1
    z = rng.rand((n, P), dev) * 4 + 1
    w, h, l = _rand_dims(rng, n, P, dev)
    return _finish(torch.stack([x, y, z, w, h, l], 2), _U.randn_orthobasis_torch(P, n, rng), gt_cubes)


def _xy_normalised(reference_box, im_shape, P):
    b = reference_box.tensor
    widths, heights = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    x_min, x_max = b[:, 0] + widths / 4, b[:, 2] - widths / 4
    y_min, y_max = b[:, 1] + heights / 4, b[:, 3] - heights / 4
    xt = pixel_to_normalised_space([x_min, x_max], [im_shape[0], im_shape[0]], [3, 3])
    yt = pixel_to_normalised_space([y_min, y_max], [im_shape[1], im_shape[1]], [2, 2])
    return _U.vectorized_linspace(xt[:, 0], xt[:, 1], P), _U.vectorized_linspace(yt[:, 0], yt[:, 1], P)


def propose_xy_patch(reference_box, depth_image, priors, im_shape, K, number_of_proposals=1, gt_cubes=None,
                     ground_normal=None, rng=None):
    """proposals.py:47-91: x / y on a line through the inner half of the 2D box (normalised image space), the rest uniform"""
    rng = rng or _U.Draws()
    n, P, dev = len(reference_box), number_of_proposals, reference_box.device
    x, y = _xy_normalised(reference_box, im_shape, P)
    z = rng.rand((n, P), dev) * 4 + 1
    w, h, l = _rand_dims(rng, n, P, dev)
    return _finish(torch.stack([x, y, z, w, h, l], 2), _U.randn_orthobasis_torch(P, n, rng), gt_cubes)


def propose_z(reference_box, depth_image, priors, im_shape, K, number_of_proposals=1, gt_cubes=None, ground_normal=None,
              rng=None):
    """proposals.py:93-135: as propose_xy_patch, z on a line between the 10 % and 90 % depth quantiles of the box's patch"""
    rng = rng or _U.Draws()
    n, P = len(reference_box), number_of_proposals
    x, y = _xy_normalised(reference_box, im_shape, P)
    b = reference_box.tensor
    z = torch.zeros_like(x)
    for i in range(n):
        patch = depth_image[int(b[i, 1]):int(b[i, 3]), int(b[i, 0]):int(b[i, 2])]
        q = torch.quantile(patch, torch.tensor([0.1, 0.9], device=patch.device), dim=None)
        z[i] = torch.linspace(float(q[0]), float(q[1]), P)
    w, h, l = _rand_dims(rng, n, P, x.device)
    return _finish(torch.stack([x, y, z, w, h, l], 2), _U.randn_orthobasis_torch(P, n, rng), gt_cubes)


def _depth_rays(reference_box, depth_image, K, P):
    """the range-to-xyz part shared by the depth-based samplers (proposals.py:140-164 = :338-375 of `propose`)"""
    b = reference_box.tensor
    widths, heights = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    xg = _U.vectorized_linspace(b[:, 0] + widths / 4, b[:, 2] - widths / 4, P).long()
    yg = _U.vectorized_linspace(b[:, 1] + heights / 4, b[:, 3] - heights / 4, P).long()
    d = depth_image[yg, xg]
    ox, oy = xg - K[0, 2].repeat(P), yg - K[1, 2].repeat(P)
    adj = K[0, 0].repeat(P)
    angle_x = torch.atan2(ox, adj)
    angle_d = torch.atan2(oy, torch.sqrt(ox ** 2 + adj ** 2))
    y = d * torch.sin(angle_d)
    dx = torch.sqrt(d ** 2 - y ** 2)
    x = dx * torch.sin(angle_x)
    return x, y, torch.sqrt(dx ** 2 - x ** 2)


def _finish_center(x, y, z_tmp, l, P, rng):
    snr = _U.sample_normal_in_range
    xs = snr(lin_fun(torch.median(x, dim=1).values, (1.15, 0)), torch.std(x, dim=1) * 1.2, P, rng=rng)
    ys = snr(lin_fun(torch.median(y, dim=1).values, (1.1, 0)), torch.std(y, dim=1) * 0.8, P, rng=rng)
    z = z_tmp + l / 2
    zs = snr(lin_fun(torch.median(z, dim=1).values, (0.85, 0.35)), torch.std(z, dim=1) * 1.2, P, rng=rng)
    return xs, ys, zs


def propose_random_dim(reference_box, depth_image, priors, im_shape, K, number_of_proposals=1, gt_cubes=None,
                       ground_normal=None, rng=None):
    """proposals.py:137-197: centre from the depth rays like `propose`, uniform dimensions, random basis"""
    rng = rng or _U.Draws()
    n, P, dev = len(reference_box), number_of_proposals, reference_box.device
    x, y, z_tmp = _depth_rays(reference_box, depth_image, K, P)
    w, h, l = _rand_dims(rng, n, P, dev)
    x, y, z = _finish_center(x, y, z_tmp, l, P, rng)
    return _finish(torch.stack([x, y, z, w, h, l], 2), _U.randn_orthobasis_torch(P, n, rng), gt_cubes)


def propose_aspect_ratio(reference_box, depth_image, priors, im_shape, K, number_of_proposals=1, gt_cubes=None,
                         ground_normal=None, rng=None):
    """proposals.py:199-270: uniform width; height and length = width x one of seven aspect ratios drawn per object"""
    rng = rng or _U.Draws()
    n, P, dev = len(reference_box), number_of_proposals, reference_box.device
    x, y, z_tmp = _depth_rays(reference_box, depth_image, K, P)
    w = rescale_interval(rng.rand((n, P), dev), MIN_PROP_S, 2)
    ratios = [0.33, 0.66, 1, 1.33, 1.67, 2, 3]
    h, l = torch.zeros_like(w), torch.zeros_like(w)
    for i in range(n):
        r1, r2 = int(rng.randperm(len(ratios))[0]), int(rng.randperm(len(ratios))[0])
        h[i] = w[i] * ratios[r1]
        l[i] = w[i] * ratios[r2]
    x, y, z = _finish_center(x, y, z_tmp, l, P, rng)
    return _finish(torch.stack([x, y, z, w, h, l], 2), _U.randn_orthobasis_torch(P, n, rng), gt_cubes)


def propose_random_rotation(reference_box, depth_image, priors, im_shape, K, number_of_proposals=1, gt_cubes=None,
                            ground_normal=None, rng=None):
    """proposals.py:272-336 (= `propose` without a ground normal, :398-400): centre and prior-driven dimensions as in
    `propose`, orientation from a random orthonormal basis instead of the 36 yaws about the ground normal"""
    rng = rng or _U.Draws()
    n, P, dev = len(reference_box), number_of_proposals, reference_box.device
    x, y, z_tmp = _depth_rays(reference_box, depth_image, K, P)
    mu, sg = priors[0].to(dev), priors[1].to(dev)
    snr = _U.sample_normal_in_range
    w = snr(mu[:, 0], sg[:, 0], P, MIN_PROP_S, mu[:, 0] + 2 * sg[:, 0], rng=rng)
    h = snr(mu[:, 1], sg[:, 1] * 1.1, P, MIN_PROP_S, mu[:, 1] + 2.2 * sg[:, 1], rng=rng)
    l = snr(mu[:, 2], sg[:, 2], P, MIN_PROP_S, mu[:, 2] + 2 * sg[:, 2], rng=rng)
    x, y, z = _finish_center(x, y, z_tmp, l, P, rng)
    ranges = None
    if gt_cubes is not None:
        pi = torch.full((gt_cubes.num_instances,), torch.pi, device=dev)
        ranges = torch.stack([torch.std(x, dim=1) * 1.2, torch.std(y, dim=1) * 0.8, torch.std(z, dim=1) * 1.2, sg[:, 0],
                              sg[:, 1] * 1.1, sg[:, 2], pi, pi, pi], dim=1).cpu().numpy()
    return _finish(torch.stack([x, y, z, w, h, l], 2), _U.randn_orthobasis_torch(P, n, rng), gt_cubes, ranges)


def statistics(gt_cubes, x, y, z, w, h, l):
    """proposals.py:427-447: where the ground-truth value lies inside the sampled range of each of the nine cube
    parameters, per object: (n,9) = [x,y,z,w,h,l, rx,ry,rz] (Euler angles against [0,pi], [0,pi/2], [0,pi])"""
    import numpy as np
    from ...cubercnn.util import math_util as util
    n = gt_cubes.num_instances
    stats = torch.zeros((n, 9))
    g = _U.gt_in_norm_range
    for i in range(n):
        gt = gt_cubes[i].tensor[0, 0]
        s6 = [g([torch.min(v[i]), torch.max(v[i])], gt[k]) for k, v in enumerate((x, y, z, w, h, l))]
        ang = util.mat2euler(gt[-9:].reshape((3, 3)).cpu().numpy())
        s3 = [g(torch.tensor([0, np.pi]), torch.tensor(ang[0])), g(torch.tensor([0, np.pi / 2]), torch.tensor(ang[1])),
              g(torch.tensor([0, np.pi]), torch.tensor(ang[2]))]
        stats[i] = torch.tensor([float(v) for v in s6 + s3])
    return stats


PROPOSAL_FUNCTIONS = {"propose": propose, "random": propose_random, "xy": propose_xy_patch, "z": propose_z,
                      "dim": propose_random_dim, "rotation": propose_random_rotation, "aspect": propose_aspect_ratio}
