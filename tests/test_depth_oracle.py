"""oracle/depth_ref.py (the CPU restatement used by bench.py's cpu_baseline of the depth workload) against the output of the
REFERENCE's own DepthAnythingV2 (tests/golden/depth_anything_vits.npz, made by tests/golden/make_golden_depth.py): same
seeded weights, same input, float32 on both sides."""
import importlib
import os

import numpy as np
import torch

from oracle import depth_ref

dav2 = importlib.import_module("3dod_amd.depth_anything_v2")
syn = importlib.import_module("3dod_amd.synthetic")
GOLD = os.path.join(os.path.dirname(__file__), "golden", "depth_anything_vits.npz")
CFG = dict(encoder="vits", features=64, out_channels=[64, 128, 256, 256], max_depth=20.0)


def test_depth_oracle_matches_reference_output():
    rec = np.load(GOLD)
    seed, shape = int(rec["seed"]), tuple(int(v) for v in rec["shape"])
    sd = syn.seeded_state_dict(dav2.DepthAnythingV2(**CFG), seed)
    x = torch.randn(shape, generator=torch.Generator().manual_seed(seed + 1))
    torch.set_num_threads(4)
    with torch.no_grad():
        depth = depth_ref.forward(sd, x, "vits", 20.0).numpy()
        feats, ph, pw = depth_ref.encoder_features({k: v.float() for k, v in sd.items()}, x, "vits")
    assert depth.shape == rec["depth"].shape
    np.testing.assert_allclose(depth, rec["depth"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(feats[3][0][:, :8, :32].numpy(), rec["feat3"], rtol=0, atol=2e-4)
    for (f, _), mu, sd_ in zip(feats, rec["feat_mean"], rec["feat_std"]):
        assert abs(float(f.mean()) - mu) < 1e-4 and abs(float(f.std()) - sd_) < 1e-4
