"""cubes_to_box -- ProposalNetwork/utils/conversions.py:25-48."""
from ...d2lite import Boxes


def cubes_to_box(cubes, K, im_shape):
    """min/max of the projected, clamped corners -> list of N Boxes (P,4)."""
    boxes = cubes._project(K, im_shape, ("boxes",))["boxes"]
    return [Boxes(boxes[i]) for i in range(cubes.num_instances)]
