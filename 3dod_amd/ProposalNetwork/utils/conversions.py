"""cubes_to_box -- ProposalNetwork/utils/conversions.py:25-48."""
from ...d2lite import Boxes


def cubes_to_box(cubes, K, im_shape):
    """min/max of the projected, clamped corners -> list of N Boxes (P,4)."""
    boxes = cubes._project(K, im_shape, ("boxes",))["boxes"]
    return [Boxes(boxes[i]) for i in range(cubes.num_instances)]


def pixel_to_normalised_space(pixel_coord, im_shape, norm_shape):
    """conversions.py:50-67: list of N coordinate tensors -> (len, N) tensor, column i shifted by half of im_shape[i] and
    scaled by norm_shape[i] / im_shape[i]"""
    import torch
    new = torch.stack(pixel_coord, dim=1).to(torch.float32)
    for i in range(new.size(1)):
        new[:, i] -= 0.5 * im_shape[i]
        new[:, i] *= norm_shape[i] / im_shape[i]
    return new
