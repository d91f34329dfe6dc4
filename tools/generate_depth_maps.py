#!/usr/bin/env python
"""Writes datasets/depth_maps/<image id>.npz {'depth': (H,W) float32 metres} for every image of the given Omni3D json
files with the Depth-Anything-V2 metric model on the GPU -- the counterpart of the reference's
cubercnn/data/generate_depth_maps.py (model_configs :13-17, loop :78-82).  The weak losses read these files through
DatasetMapper3D.

    python tools/generate_depth_maps.py --weights depth/checkpoints/depth_anything_v2_metric_hypersim_vitl.pth \\
        --datasets SUNRGBD_train SUNRGBD_val [--encoder vitl --max-depth 20 --root datasets]
"""
import argparse
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODEL_CONFIGS = {
    'vits': {'encoder': 'vits', 'features': 64, 'out_channels': [48, 96, 192, 384]},
    'vitb': {'encoder': 'vitb', 'features': 128, 'out_channels': [96, 192, 384, 768]},
    'vitl': {'encoder': 'vitl', 'features': 256, 'out_channels': [256, 512, 1024, 1024]},
}


def build_model(encoder='vitl', max_depth=20.0, weights=None, device='cuda:0', **overrides):
    dav2 = importlib.import_module("3dod_amd.depth_anything_v2")
    model = dav2.DepthAnythingV2(**{**MODEL_CONFIGS[encoder], **overrides, 'max_depth': max_depth})
    if weights:
        model.load_state_dict(torch.load(weights, map_location='cpu', weights_only=True))
    return model.to(device).eval()


def generate(model, json_files, root='datasets', out_dir=None, skip_existing=True):
    data = importlib.import_module("3dod_amd.d2lite.data")
    out_dir = out_dir or os.path.join(root, 'depth_maps')
    os.makedirs(out_dir, exist_ok=True)
    n = 0
    for jf in json_files:
        with open(jf) as f:
            images = json.load(f)['images']
        for info in images:
            dst = os.path.join(out_dir, f"{info['id']}.npz")
            if skip_existing and os.path.exists(dst):
                continue
            img = data.read_image(os.path.join(root, info['file_path']), format='BGR')
            depth = model.infer_image(np.ascontiguousarray(img))
            np.savez_compressed(dst, depth=depth.astype(np.float32))
            n += 1
    return n


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--encoder', default='vitl', choices=sorted(MODEL_CONFIGS))
    ap.add_argument('--max-depth', type=float, default=20.0, help='20 for the indoor (hypersim) model, 80 for vkitti')
    ap.add_argument('--weights', required=True)
    ap.add_argument('--datasets', nargs='+', required=True)
    ap.add_argument('--root', default='datasets')
    a = ap.parse_args()
    m = build_model(a.encoder, a.max_depth, a.weights)
    files = [os.path.join(a.root, 'Omni3D', d + '.json') for d in a.datasets]
    print('wrote', generate(m, files, a.root), 'depth maps')
