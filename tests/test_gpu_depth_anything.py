"""Depth-Anything-V2 forward (DINOv2 ViT + DPT head) on the HIP kernels against the output of the REFERENCE's float32
model with the same seeded weights (tests/golden/make_golden_depth.py).  Activations are bf16 end to end here, so the
tolerance is that of bf16 storage through 12 transformer blocks and the DPT head, stated below."""
import importlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
dav2 = importlib.import_module("3dod_amd.depth_anything_v2")
syn = importlib.import_module("3dod_amd.synthetic")
GOLD = os.path.join(os.path.dirname(__file__), "golden", "depth_anything_vits.npz")
CFG = dict(encoder="vits", features=64, out_channels=[64, 128, 256, 256], max_depth=20.0)


def build(seed, dev):
    m = dav2.DepthAnythingV2(**CFG)
    m.load_state_dict(syn.seeded_state_dict(m, seed))
    return m.to(dev).eval()


def test_matches_reference_forward():
    dev = torch.device("cuda:0")
    rec = np.load(GOLD)
    seed, shape = int(rec["seed"]), tuple(int(v) for v in rec["shape"])
    model = build(seed, dev)
    x = torch.randn(shape, generator=torch.Generator().manual_seed(seed + 1))
    depth = model(x.to(dev)).cpu().numpy()
    want = rec["depth"]
    assert depth.shape == want.shape and depth.dtype == np.float32 and np.isfinite(depth).all()
    err = np.abs(depth - want)
    # depth range of this case: 2.1 .. 15.6 m (std 2.1); bf16 activations: mean error < 0.5 % of the range, max < 3 %
    assert err.mean() < 0.06 and err.max() < 0.45, (err.mean(), err.max())
    assert np.corrcoef(depth.ravel(), want.ravel())[0, 1] > 0.999
    # the encoder alone: last tapped layer, a slice of its patch tokens
    feats = model.pretrained.get_intermediate_layers(x.to(dev), model.intermediate_layer_idx["vits"], return_class_token=True)
    f3 = feats[3][0][:, :8, :32].float().cpu().numpy()
    assert np.abs(f3 - rec["feat3"]).mean() < 0.03 and np.abs(f3 - rec["feat3"]).max() < 0.25
    for f, mu, sd in zip(feats, rec["feat_mean"], rec["feat_std"]):
        assert abs(float(f[0].float().mean()) - mu) < 0.01 and abs(float(f[0].float().std()) - sd) < 0.02


def test_other_input_sizes_and_state_dict_keys():
    dev = torch.device("cuda:0")
    model = build(1, dev)
    for hw in ((14, 14), (518, 518), (70, 210)):
        d = model(torch.randn(1, 3, *hw, device=dev))
        assert d.shape == (1, *hw) and torch.isfinite(d).all() and float(d.min()) >= 0 and float(d.max()) <= 20.0
    keys = list(model.state_dict().keys())
    assert "pretrained.blocks.11.attn.qkv.weight" in keys and "depth_head.scratch.refinenet4.resConfUnit2.conv2.bias" in keys
    assert "depth_head.resize_layers.0.weight" in keys and "depth_head.scratch.output_conv2.2.weight" in keys
    with pytest.raises(RuntimeError):
        model(torch.randn(1, 3, 28, 28))          # CPU tensor: no CPU path


def test_infer_image_and_depth_map_tool(tmp_path):
    """raw BGR image -> metric depth at the image's size; tools/generate_depth_maps.py writes the .npz files the weak
    losses' data path reads"""
    import sys
    dev = torch.device("cuda:0")
    assert dav2.DepthAnythingV2._net_size(480, 640) == (518, 686) and dav2.DepthAnythingV2._net_size(518, 518) == (518, 518)
    assert dav2.DepthAnythingV2._net_size(375, 1242) == (518, 1722)
    model = build(2, dev)
    img = np.random.default_rng(0).integers(0, 256, (60, 90, 3), dtype=np.uint8)
    d = model.infer_image(img, input_size=70)
    assert d.shape == (60, 90) and d.dtype == np.float32 and np.isfinite(d).all() and d.min() >= 0 and d.max() <= 20
    root = tmp_path / "datasets"
    root.mkdir()
    jf = syn.make_omni3d_dataset(str(root), name="Synth_train", n_images=3, seed=1, with_maps=False)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    tool = importlib.import_module("generate_depth_maps")
    model.infer_image.__func__.__defaults__ = (70,)            # small network input for the test
    try:
        n = tool.generate(model, [jf], root=str(root))
    finally:
        model.infer_image.__func__.__defaults__ = (518,)
    assert n == 3 and tool.generate(model, [jf], root=str(root)) == 0        # second run: nothing left to do
    import json
    for info in json.load(open(jf))["images"]:
        with np.load(root / "depth_maps" / f"{info['id']}.npz") as f:
            assert f["depth"].shape == (info["height"], info["width"]) and f["depth"].dtype == np.float32
