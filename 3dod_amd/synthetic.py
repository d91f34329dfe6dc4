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


def add_scene_maps(batch, seed, ground_every=1):
    """adds the inputs of the weakly supervised model to a `make_batch` batch: `depth_map` (h,w) float32 = a ground plane
    1.5 m below the camera with a back wall at 8 m plus 1 cm noise, `ground_map` (h,w) bool (None for every image whose
    index is not a multiple of `ground_every`, like an image without ground segmentation)."""
    g = torch.Generator().manual_seed(seed)
    for i, d in enumerate(batch):
        h, w = d["image"].shape[-2:]
        f = d["K"][0][0]
        v = torch.arange(h, dtype=torch.float32).view(-1, 1).expand(h, w)
        z = torch.where(v > h / 2 + 8, 1.5 * f / (v - h / 2).clamp(min=1.0), torch.full_like(v, 8.0)).clamp(max=8.0)
        d["depth_map"] = (z + torch.randn(h, w, generator=g) * 0.01).contiguous()
        d["ground_map"] = (v > h / 2 + 40).contiguous() if i % ground_every == 0 else None
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


_SYNTH_CATEGORY_IDS = {n: 3 * i + 2 for i, n in enumerate(
    ("bed", "car", "chair", "lamp", "sofa", "table", "truck", "dontcare"))}


def make_omni3d_dataset(root, name="Synth_train", n_images=6, seed=0, sizes=((480, 640), (512, 512), (640, 400)),
                        categories=("bed", "car", "chair", "sofa", "table", "truck"), dataset_id=90, source="synthetic",
                        first_image_id=1000, with_maps=True, extra_category=None):
    """Writes a small dataset in the Omni3D wire format (DATA.md:140-200 of the reference) under `root`:

        <root>/Omni3D/<name>.json          annotations
        <root>/Omni3D/stats.json           global category table used by register_and_store_model_metadata
        <root>/<name>/images/<id>.png      images (RGB on disk)
        <root>/depth_maps/<id>.npz {'depth'}   <root>/ground_maps/<id>.npz {'mask'}   (every second image has no ground)
        <root>/no_ground_idx.csv           ids of the images without a ground map

    Objects get consistent geometry (corners projected with K), plus a controlled mix of the cases the filters look
    at: behind camera, invalid 3D, tiny dimensions, far depth, zero lidar / segmentation points, large depth error,
    heavy truncation, low visibility, missing tight / truncated boxes.  Returns the json path."""
    import json
    import os
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, "Omni3D"), exist_ok=True)
    os.makedirs(os.path.join(root, name, "images"), exist_ok=True)
    if with_maps:
        os.makedirs(os.path.join(root, "depth_maps"), exist_ok=True)
        os.makedirs(os.path.join(root, "ground_maps"), exist_ok=True)
    cat_names = list(categories) + ([extra_category] if extra_category else [])
    # category ids are global across files (as in Omni3D): fixed by name, not by position in this file
    cats = [{"id": _SYNTH_CATEGORY_IDS.get(n, 900 + i), "name": n, "supercategory": "object"} for i, n in enumerate(cat_names)]
    cats.sort(key=lambda c: c["id"])
    images, annos = [], []
    aid = first_image_id * 100
    corner_signs = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1],
                             [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], dtype=np.float64)
    for i in range(n_images):
        h, w = sizes[i % len(sizes)]
        iid = first_image_id + i
        f = float(rng.uniform(400, 800))
        K = [[f, 0.0, w / 2], [0.0, f, h / 2], [0.0, 0.0, 1.0]]
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        rel = os.path.join(name, "images", f"{iid}.png")
        Image.fromarray(img).save(os.path.join(root, rel))
        images.append({"id": iid, "dataset_id": dataset_id, "width": w, "height": h, "file_path": rel, "K": K,
                       "src_90_rotate": 0, "src_flagged": False})
        if with_maps:
            np.savez_compressed(os.path.join(root, "depth_maps", f"{iid}.npz"),
                                depth=rng.uniform(1, 8, (h // 2, w // 2)).astype(np.float32))
            if i % 2 == 0:
                np.savez_compressed(os.path.join(root, "ground_maps", f"{iid}.npz"),
                                    mask=(rng.uniform(size=(h // 2, w // 2)) > 0.5))
        for j in range(int(rng.integers(3, 9))):
            c = cats[int(rng.integers(0, len(cats)))]
            z = float(rng.uniform(1.5, 9))
            u, v = float(rng.uniform(0.1 * w, 0.9 * w)), float(rng.uniform(0.15 * h, 0.85 * h))
            center = [z * (u - w / 2) / f, z * (v - h / 2) / f, z]
            dims = (np.exp(rng.normal(size=3) * 0.35) * 0.7).tolist()          # (w, h, l)
            yaw = float(rng.uniform(-np.pi, np.pi))
            R = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
            half = np.array([dims[2], dims[1], dims[0]]) / 2                   # local axes: l along x, h along y, w along z
            corners = (R @ (corner_signs * half).T).T + np.array(center)
            case = int(rng.integers(0, 14))                                    # 0-4: plain objects
            behind = case == 5
            if behind:
                center[2] = -abs(center[2])
                corners[:, 2] = -np.abs(corners[:, 2])
            pz = np.where(np.abs(corners[:, 2]) < 1e-3, 1e-3, corners[:, 2])
            pu, pv = f * corners[:, 0] / pz + w / 2, f * corners[:, 1] / pz + h / 2
            proj = [float(pu.min()), float(pv.min()), float(pu.max()), float(pv.max())]
            trunc = [float(np.clip(proj[0], 0, w)), float(np.clip(proj[1], 0, h)),
                     float(np.clip(proj[2], 0, w)), float(np.clip(proj[3], 0, h))]
            area_p = max((proj[2] - proj[0]) * (proj[3] - proj[1]), 1e-6)
            truncation = float(1 - (trunc[2] - trunc[0]) * (trunc[3] - trunc[1]) / area_p)
            a = {"id": aid, "image_id": iid, "dataset_id": dataset_id, "category_id": c["id"], "category_name": c["name"],
                 "valid3D": case != 6, "bbox2D_tight": [-1, -1, -1, -1] if j % 3 else [t + 1.5 for t in trunc],
                 "bbox2D_proj": proj, "bbox2D_trunc": [-1, -1, -1, -1] if case == 13 else trunc,
                 "bbox3D_cam": corners.tolist(), "center_cam": center, "dimensions": dims, "R_cam": R.tolist(),
                 "behind_camera": behind, "visibility": 0.005 if case == 7 else float(rng.uniform(0.3, 1.0)),
                 "truncation": 0.995 if case == 8 else (-1 if case == 9 else min(truncation, 0.9)),
                 "segmentation_pts": 0 if case == 10 else int(rng.integers(10, 5000)),
                 "lidar_pts": 0 if case == 11 else (-1 if j % 2 else int(rng.integers(5, 900))),
                 "depth_error": 0.8 if case == 12 else (-1 if j % 2 else float(rng.uniform(0, 0.3)))}
            if case == 4:
                a["dimensions"] = [dims[0], 0.005, dims[2]]
            annos.append(a)
            aid += 1
    if with_maps:                                       # ids of images without a ground map (datasets.py:151 reads this)
        import csv
        csv_path = os.path.join(root, "no_ground_idx.csv")
        old_rows = []
        if os.path.exists(csv_path):
            with open(csv_path) as fcsv:
                old_rows = [r for r in csv.reader(fcsv)][1:]
        with open(csv_path, "w", newline="") as fcsv:
            wr = csv.writer(fcsv)
            wr.writerow(["img_id"])
            wr.writerows(old_rows + [[first_image_id + i] for i in range(n_images) if i % 2])
    info = {"id": dataset_id, "source": source, "name": name, "split": "Train", "version": "0.1", "url": ""}
    path = os.path.join(root, "Omni3D", name + ".json")
    with open(path, "w") as fjson:
        json.dump({"info": info, "images": images, "categories": cats, "annotations": annos}, fjson)
    stats_path = os.path.join(root, "Omni3D", "stats.json")
    stats = {"n_datasets": 0, "n_ims": 0, "n_anns": 0, "categories": [], "category_names": []}
    if os.path.exists(stats_path):
        with open(stats_path) as fjson:
            stats = json.load(fjson)
    for c in cats:
        if c["name"] not in stats["category_names"]:
            stats["category_names"].append(c["name"])
            stats["categories"].append(c)
    stats["n_datasets"] += 1
    stats["n_ims"] += len(images)
    stats["n_anns"] += len(annos)
    with open(stats_path, "w") as fjson:
        json.dump(stats, fjson)
    return path


def seeded_state_dict(model, seed):
    """deterministic weights for ANY module from its state-dict keys and shapes alone (sorted keys, one generator): the
    same call on the reference's module and on this repo's module of the same architecture gives both the same weights
    without shipping a checkpoint.  Scales keep activations O(1) through deep stacks."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sorted(model.state_dict().items()):
        if not v.dtype.is_floating_point:
            out[k] = v.clone()
            continue
        r = torch.randn(v.shape, generator=g)
        name = k.rsplit(".", 1)[-1]
        if v.dim() >= 2 and name == "weight":
            fan_in = v[0].numel()
            t = r / math.sqrt(fan_in)
        elif "norm" in k and name == "weight":
            t = 1.0 + 0.1 * r
        elif name == "gamma":
            t = 1.0 + 0.1 * r
        elif name == "bias":
            t = 0.05 * r
        else:                                   # position table, class / mask tokens
            t = 0.02 * r
        out[k] = t.to(v.dtype)
    return out
