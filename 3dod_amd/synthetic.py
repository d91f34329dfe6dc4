"""Synthetic Omni3D-shaped inputs: 512x512 uint8 BGR images, K = [[f,0,256],[0,f,256],[0,0,1]]
with f ~ U(400,800), 4..16 GT objects per image with 2D boxes, gt_boxes3D = [cx2d,cy2d,z,w,h,l,X,Y,Z]
(cubercnn/data/dataset_mapper.py:258 of the reference), random yaw poses, 50 classes."""
import math

import torch

from .d2lite import Boxes, Instances


def make_batch(n_images, seed, size=512, num_classes=50, min_obj=4, max_obj=16, with_gt=True):
    g = torch.Generator().manual_seed(seed)
    batch = []
    for _ in range(n_images):
        img = torch.randint(0, 256, (3, size, size), generator=g, dtype=torch.uint8)
        f = float(torch.empty(1).uniform_(400, 800, generator=g))
        K = [[f, 0.0, size / 2], [0.0, f, size / 2], [0.0, 0.0, 1.0]]
        d = {"image": img, "height": size, "width": size, "K": K}
        if with_gt:
            G = int(torch.randint(min_obj, max_obj + 1, (1,), generator=g))
            ctr = torch.empty(G, 2).uniform_(64, size - 64, generator=g)
            wh = torch.empty(G, 2).uniform_(32, 256, generator=g)
            boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1).clamp(0, size - 1)
            z = torch.empty(G).uniform_(1, 8, generator=g)
            dims = torch.exp(torch.randn(G, 3, generator=g) * 0.4) * 0.6
            c2 = (boxes[:, :2] + boxes[:, 2:]) / 2
            X = z * (c2[:, 0] - size / 2) / f
            Y = z * (c2[:, 1] - size / 2) / f
            yaw = torch.empty(G).uniform_(-math.pi, math.pi, generator=g)
            tilt = torch.randn(G, generator=g) * 0.05
            cy, sy, ct, st = yaw.cos(), yaw.sin(), tilt.cos(), tilt.sin()
            o, l = torch.zeros(G), torch.ones(G)
            Ry = torch.stack([cy, o, sy, o, l, o, -sy, o, cy], 1).view(G, 3, 3)
            Rx = torch.stack([l, o, o, o, ct, -st, o, st, ct], 1).view(G, 3, 3)
            inst = Instances((size, size))
            inst.gt_boxes = Boxes(boxes)
            inst.gt_classes = torch.randint(0, num_classes, (G,), generator=g)
            inst.gt_boxes3D = torch.cat([c2, z[:, None], dims, X[:, None], Y[:, None], z[:, None]], 1)
            inst.gt_poses = Ry @ Rx
            d["instances"] = inst
        batch.append(d)
    return batch


def make_cfg(config_file=None, overrides=()):
    """get_cfg + get_cfg_defaults + yaml + overrides, like tools/train_net.py:335-353 of the reference."""
    import os
    from .d2lite import get_cfg
    from .cubercnn.config import get_cfg_defaults
    cfg = get_cfg()
    get_cfg_defaults(cfg)
    if config_file is None:
        config_file = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs",
                                   "Base_Omni3D.yaml")
    cfg.merge_from_file(config_file)
    cfg.merge_from_list(list(overrides))
    return cfg
