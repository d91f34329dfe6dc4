"""propose -- ProposalNetwork/proposals/proposals.py:338-424 of the reference.  The arithmetic runs in
cr_propose; this wrapper draws the random variates on the device (torch.randn / torch.randint, like the
reference's torch.normal / torch.randint) and redraws while the truncated-normal rejection rounds are exhausted
(sample_normal_in_range loops up to 10 000 rounds, utils.py:42-60)."""
import torch

from ... import geometry as geo
from ..utils.spaces import Cubes

MIN_PROP_S = 0.05
ROUNDS = 8
_rounds_hint = [ROUNDS]


def propose(reference_box, depth_image, priors, im_shape, K, number_of_proposals=1, gt_cubes=None,
            ground_normal: torch.Tensor = None, generator=None):
    """reference_box: Boxes (N,4); depth_image (H,W); priors = (mean (N,3), std (N,3)) in (w,h,l); K (3,3).
    Returns (Cubes (N,P,15), None, None) like the reference without gt_cubes."""
    if ground_normal is None:
        raise NotImplementedError("the random-orthobasis variant (ground_normal=None) is not built; BoxNet always "
                                  "passes a normal (roi_heads.py:493)")
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
    return Cubes(cubes), None, None


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
