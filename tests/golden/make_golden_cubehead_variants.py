"""Golden vector for the CubeHead with MODEL.ROI_CUBE_HEAD.SHARED_FC = False (cubercnn/modeling/roi_heads/cube_head.py:56-111,
160-178: one FC trunk per predictor): the REFERENCE's own CubeHead, seeded, with its state dict, an input and its five outputs.
Third-party stand-ins as in make_golden_dense.py (fvcore c2_xavier_fill, pytorch3d rotation_6d_to_matrix).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_cubehead_variants.py
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_dense as M  # noqa: E402  (imports the reference under the stub finder; does not run its main())

d2 = importlib.import_module("3dod_amd.d2lite")


def run(seed):
    cfg = importlib.import_module("3dod_amd.synthetic").make_cfg()
    cfg.MODEL.ROI_HEADS.NUM_CLASSES = 7
    cfg.MODEL.ROI_CUBE_HEAD.FC_DIM = 16
    cfg.MODEL.ROI_CUBE_HEAD.NUM_FC = 2
    cfg.MODEL.ROI_CUBE_HEAD.SHARED_FC = False
    C, H, W = 16, 7, 7
    torch.manual_seed(seed)
    head = M.ref_ch.CubeHead(cfg, d2.ShapeSpec(channels=C, height=H, width=W)).eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in (head.bbox_3D_center_deltas, head.bbox_3D_dims, head.bbox_3D_pose, head.bbox_3D_center_depth, head.bbox_3D_uncertainty):
            m.weight.add_(torch.randn(m.weight.shape, generator=g) * 0.05)
    x = torch.randn(11, C, H, W, generator=g)
    with torch.no_grad():
        d, z, dims, pose, unc = head(x.flatten(1))
    out = dict(x=x, deltas=d, z=z, dims=dims, pose=pose, uncert=unc, cfg=torch.tensor([7, 16, 2, C, H, W]), seed=torch.tensor(seed))
    for k, v in head.state_dict().items():
        out["sd." + k] = v
    return {"cube_" + k: v.detach().numpy() for k, v in out.items()}


if __name__ == "__main__":
    rec = run(21)
    rec["notes"] = np.array("reference CubeHead (cube_head.py:24-202) with SHARED_FC = False; stand-ins: c2_xavier_fill, rotation_6d_to_matrix")
    np.savez_compressed(os.path.join(HERE, "cubehead_nonshared.npz"), **rec)
    print(sorted(k for k in rec if k.startswith("cube_sd."))[:8], len(rec))
