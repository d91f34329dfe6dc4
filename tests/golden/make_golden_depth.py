"""Golden vector for the Depth-Anything-V2 forward: runs the REFERENCE's DepthAnythingV2
(depth/metric_depth/depth_anything_v2/dpt.py:159-189 with dinov2.py and util/blocks.py) on CPU in float32 with weights
made by `3dod_amd.synthetic.seeded_state_dict` (the test gives this repo's model the same weights the same way) and
records input seed -> output depth, plus the four intermediate encoder features' statistics.

The reference's package is pure PyTorch (cv2 / torchvision are only used by its image reader and are stubbed; xformers is
absent, so its attention is the plain softmax path): nothing third-party carries arithmetic here.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_depth.py
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _refimport  # noqa: E402

_refimport.install()
sys.path.insert(0, os.path.join(_refimport.REFERENCE, "depth", "metric_depth"))
from depth_anything_v2.dpt import DepthAnythingV2  # noqa: E402  (reference)

syn = importlib.import_module("3dod_amd.synthetic")
mine = importlib.import_module("3dod_amd.depth_anything_v2")

CFG = dict(encoder="vits", features=64, out_channels=[64, 128, 256, 256], max_depth=20.0)
SEED, SHAPE = 5, (2, 3, 98, 140)

if __name__ == "__main__":
    torch.set_num_threads(4)
    ref = DepthAnythingV2(**CFG).eval()
    my = mine.DepthAnythingV2(**CFG)
    assert list(ref.state_dict().keys()) == list(my.state_dict().keys()), "state-dict keys differ"
    assert all(a.shape == b.shape for a, b in zip(ref.state_dict().values(), my.state_dict().values()))
    ref.load_state_dict(syn.seeded_state_dict(ref, SEED))
    x = torch.randn(SHAPE, generator=torch.Generator().manual_seed(SEED + 1))
    with torch.no_grad():
        depth = ref(x)
        feats = ref.pretrained.get_intermediate_layers(x, ref.intermediate_layer_idx["vits"], return_class_token=True)
    rec = {"depth": depth.numpy(), "seed": np.array(SEED), "shape": np.array(SHAPE),
           "feat_mean": np.array([float(f[0].mean()) for f in feats]), "feat_std": np.array([float(f[0].std()) for f in feats]),
           "feat3": feats[3][0][:, :8, :32].numpy(),
           "notes": "reference DepthAnythingV2(vits, features=64, out_channels=[64,128,256,256]) float32 on CPU, weights = "
                    "seeded_state_dict(model, 5), input = randn(seed 6)"}
    np.savez_compressed(os.path.join(HERE, "depth_anything_vits.npz"), **rec)
    print("depth", depth.shape, float(depth.mean()), float(depth.std()), float(depth.min()), float(depth.max()))
    print("feat std", rec["feat_std"])
