#!/usr/bin/env python
"""Cube R-CNN on a folder of images -- the counterpart of the reference's demo/demo.py (do_test :24-163): for every image
<OUTPUT_DIR>/<name>_boxes.jpg (the detections above --threshold as projected 3D wireframes with labels) and
<name>_novel.jpg (a top-down view of the same boxes; the reference renders a mesh scene with pytorch3d), plus <name>.json
({category, score, bbox3D [X,Y,Z,w,h,l], pose 3x3, corners3D 8x3, center_2D, bbox2D}).

    python tools/demo.py --config-file configs/Base_Omni3D.yaml --input-folder datasets/demo_images \\
        --threshold 0.25 MODEL.WEIGHTS output/run1/model_final.pth OUTPUT_DIR output/demo

Intrinsics as in the reference (:63-77): --focal-length 0 means 4.0 in NDC (= 2 * image height), --principal-point empty
means the image centre.  `category_meta.json` is looked up next to the weights, then next to the config (:46).
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


def camera(h, w, focal_length=0.0, principal_point=()):
    f = focal_length if focal_length else 4.0 * h / 2
    px, py = principal_point if len(principal_point) else (w / 2, h / 2)
    return [[f, 0.0, px], [0.0, f, py], [0.0, 0.0, 1.0]]


def run(cfg, model, files, out_dir, cats, threshold=0.25, focal_length=0.0, principal_point=()):
    D = importlib.import_module("3dod_amd.d2lite.data")
    vis = importlib.import_module("3dod_amd.cubercnn.vis")
    U = importlib.import_module("3dod_amd.cubercnn.util.util")
    os.makedirs(out_dir, exist_ok=True)
    aug = D.AugmentationList([D.ResizeShortestEdge(cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST, "choice")])
    model.eval()
    written = []
    for path in files:
        try:
            im = D.read_image(path, format="BGR")
        except Exception:
            continue                                   # not an image (the reference skips what imread cannot open)
        h, w = im.shape[:2]
        inp = D.AugInput(np.ascontiguousarray(im))
        aug(inp)
        batched = [{"image": torch.as_tensor(np.ascontiguousarray(inp.image.transpose(2, 0, 1))), "height": h, "width": w,
                    "K": camera(h, w, focal_length, principal_point)}]
        with torch.no_grad():
            dets = model(batched)[0]["instances"].to("cpu")
        out = []
        for k in range(len(dets)):
            score = float(dets.scores[k])
            if score < threshold:
                continue
            out.append({"category": cats[int(dets.pred_classes[k])], "score": score,
                        "bbox3D": dets.pred_center_cam[k].tolist() + dets.pred_dimensions[k].tolist(),
                        "pose": dets.pred_pose[k].tolist(), "corners3D": dets.pred_bbox3D[k].tolist(),
                        "center_2D": dets.pred_center_2D[k].tolist(), "bbox2D": dets.pred_boxes.tensor[k].tolist()})
        name = os.path.splitext(os.path.basename(path))[0]
        dst = os.path.join(out_dir, name + ".json")
        with open(dst, "w") as f:
            json.dump({"file": path, "K": batched[0]["K"], "detections": out}, f)
        written.append(dst)
        K = batched[0]["K"]
        if out:                                        # demo.py:116-142
            meshes = [U.mesh_cuboid(o["bbox3D"], o["pose"], color=[c / 255.0 for c in U.get_color(i)]) for i, o in enumerate(out)]
            text = ["{} {:.2f}".format(o["category"], o["score"]) for o in out]
            im_drawn, im_topdown, _ = vis.draw_scene_view(im, K, meshes, text=text, scale=im.shape[0], blend_weight=0.5,
                                                          blend_weight_overlay=0.85)
            U.imwrite(im_drawn, os.path.join(out_dir, name + "_boxes.jpg"))
            U.imwrite(im_topdown, os.path.join(out_dir, name + "_novel.jpg"))
        else:
            U.imwrite(im, os.path.join(out_dir, name + "_boxes.jpg"))
        print("File: {} with {} dets".format(name, len(out)))
    return written


def main(args):
    syn = importlib.import_module("3dod_amd.synthetic")
    modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
    solver = importlib.import_module("3dod_amd.cubercnn.solver")
    cfg = syn.make_cfg(args.config_file, overrides=args.opts)
    if "MODEL.DEVICE" not in args.opts:
        cfg.MODEL.DEVICE = "cuda:0"
    weights = cfg.MODEL.get("WEIGHTS", "")
    meta = None
    for d in (os.path.dirname(weights) if weights else None, os.path.dirname(args.config_file), cfg.OUTPUT_DIR):
        if d and os.path.exists(os.path.join(d, "category_meta.json")):
            with open(os.path.join(d, "category_meta.json")) as f:
                meta = json.load(f)
            break
    if meta is None:
        raise FileNotFoundError("category_meta.json not found next to the weights, the config or in OUTPUT_DIR")
    cats = meta["thing_classes"]
    cfg.MODEL.ROI_HEADS.NUM_CLASSES = len(cats)
    model = modeling.build_model(cfg)
    solver.Checkpointer(model, None).load(weights, checkpointables=[])
    files = sorted(os.path.join(args.input_folder, f) for f in os.listdir(args.input_folder))
    return run(cfg, model, files, cfg.OUTPUT_DIR, cats, args.threshold, args.focal_length, args.principal_point)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-file", default=os.path.join(ROOT, "configs", "Base_Omni3D.yaml"))
    ap.add_argument("--input-folder", required=True)
    ap.add_argument("--threshold", type=float, default=0.25)
    ap.add_argument("--focal-length", type=float, default=0.0)
    ap.add_argument("--principal-point", type=float, nargs=2, default=[])
    ap.add_argument("opts", nargs=argparse.REMAINDER, default=[])
    a = ap.parse_args()

    def _val(v):
        import ast
        try:
            return ast.literal_eval(v)
        except Exception:
            return v
    a.opts = [(_val(v) if i % 2 else v) for i, v in enumerate(a.opts)]
    main(a)
