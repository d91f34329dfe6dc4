"""Golden vectors for the Omni3D data path: runs the REFERENCE's own functions on a synthetic dataset written by
`3dod_amd.synthetic.make_omni3d_dataset` (seeded; the test re-creates the same files) and records their outputs.

Reference functions executed (from /root/reference, build container only):
    cubercnn/data/datasets.py       is_ignore, Omni3D.__init__ (filtering), load_omni3d_json
    cubercnn/data/dataset_mapper.py transform_instance_annotations, annotations_to_instances
    cubercnn/data/build.py          repeat_factors_from_category_frequency
    cubercnn/util/math_util.py      approx_eval_resolution, compute_priors (1 and 3 cluster bins)

Third-party symbols they touch are stood in by this repo's d2lite restatements (pycocotools / detectron2 / fvcore are
not installed): the COCO index, BoxMode, MetadataCatalog, TransformList / HFlipTransform / ResizeTransform, Boxes,
Instances, Keypoints -- results that flow through those are "parity unpinned" at that boundary; the filtering rules,
record layout, pose mirroring, gt_boxes3D packing, repeat factors and prior statistics are the reference's code.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_data.py
"""
import importlib
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _refimport  # noqa: E402

d2data = importlib.import_module("3dod_amd.d2lite.data")
d2 = importlib.import_module("3dod_amd.d2lite")
my_ds = importlib.import_module("3dod_amd.cubercnn.data.datasets")
syn = importlib.import_module("3dod_amd.synthetic")

# the COCO base class must exist before the reference defines `class Omni3D(COCO)`
pc = types.ModuleType("pycocotools"); pc.__path__ = []
pcc = types.ModuleType("pycocotools.coco"); pcc.COCO = my_ds._CocoIndex
sys.modules["pycocotools"], sys.modules["pycocotools.coco"] = pc, pcc
_refimport.install()

import cubercnn.data.datasets as ref_ds            # noqa: E402  (reference)
import cubercnn.data.dataset_mapper as ref_dm      # noqa: E402
import cubercnn.data.build as ref_build            # noqa: E402
from cubercnn.util import math_util as ref_math    # noqa: E402


class _Timer:
    def seconds(self):
        return 0.0


ref_ds.BoxMode = d2data.BoxMode
ref_ds.MetadataCatalog = d2data.MetadataCatalog
ref_ds.DatasetCatalog = d2data.DatasetCatalog
ref_ds.Timer = _Timer
ref_ds.PathManager = types.SimpleNamespace(get_local_path=lambda p: p)
ref_dm.BoxMode = d2data.BoxMode
ref_dm.T = types.SimpleNamespace(TransformList=d2data.TransformList, HFlipTransform=d2data.HFlipTransform)
ref_dm.Instances, ref_dm.Boxes, ref_dm.Keypoints = d2.Instances, d2.Boxes, d2data.Keypoints
ref_math.BoxMode = d2data.BoxMode
ref_math.MetadataCatalog = d2data.MetadataCatalog

GEN = {"name": "Synth_train", "n_images": 12, "seed": 7}
GEN2 = {"name": "Synth_b_train", "n_images": 5, "seed": 8, "dataset_id": 91, "source": "synthetic_b",
        "first_image_id": 5000, "categories": ["car", "chair", "lamp"], "extra_category": "dontcare"}
SETTINGS = {
    "default": dict(category_names=["bed", "car", "chair", "sofa", "table"], ignore_names=["dontcare"],
                    truncation_thres=0.33, visibility_thres=0.33, min_height_thres=0.05, max_height_thres=1.50,
                    modal_2D_boxes=False, trunc_2D_boxes=True, max_depth=1e8),
    "modal_nearonly": dict(category_names=["car", "chair", "truck"], ignore_names=[], truncation_thres=0.99,
                           visibility_thres=0.01, min_height_thres=0.0, max_height_thres=1.50,
                           modal_2D_boxes=True, trunc_2D_boxes=False, max_depth=6.0),
    "all_categories": dict(category_names=[], ignore_names=[], truncation_thres=0.5, visibility_thres=0.2,
                           min_height_thres=0.0, max_height_thres=0.6, modal_2D_boxes=False, trunc_2D_boxes=False,
                           max_depth=1e8),
}


def tolist(x):
    if isinstance(x, torch.Tensor):
        return x.tolist()
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.floating, np.integer, np.bool_)):
        return x.item()
    if isinstance(x, dict):
        return {str(k): tolist(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [tolist(v) for v in x]
    return x


def main():
    out = {"gen": [GEN, GEN2], "settings": SETTINGS, "cases": {}}
    work = tempfile.mkdtemp()
    root = os.path.join(work, "datasets")
    os.makedirs(root)
    p1 = syn.make_omni3d_dataset(root, **GEN)
    p2 = syn.make_omni3d_dataset(root, **GEN2)
    os.chdir(work)                                   # the reference addresses 'datasets/...' relative to the cwd
    rel = [os.path.relpath(p, work) for p in (p1, p2)]

    for sname, fs in SETTINGS.items():
        case = {}
        raw = [json.load(open(p)) for p in rel]
        heights = {im["id"]: im["height"] for r in raw for im in r["images"]}
        case["is_ignore"] = {str(a["id"]): bool(ref_ds.is_ignore(a, fs, heights[a["image_id"]]))
                             for r in raw for a in r["annotations"]}

        fs_run = json.loads(json.dumps(fs))          # Omni3D fills category_names in place when empty
        omni = ref_ds.Omni3D(rel, filter_settings=fs_run)
        case["omni3d"] = {
            "category_names_after": fs_run["category_names"],
            "categories": [c["name"] for c in omni.dataset["categories"]],
            "kept": [{k: a[k] for k in ("id", "area", "ignore", "bbox", "depth")} for a in omni.dataset["annotations"]],
            "known_category_ids": [i["known_category_ids"] for i in omni.dataset["info"]],
        }

        # model metadata the loader maps category ids with (register_and_store_model_metadata, fresh output dir)
        d2data.MetadataCatalog.pop("omni3d_model", None)
        outdir = tempfile.mkdtemp()
        ref_ds.register_and_store_model_metadata(omni, outdir, fs_run)
        meta = d2data.MetadataCatalog.get("omni3d_model")
        case["thing_classes"] = list(meta.thing_classes)
        case["id_map"] = {str(k): v for k, v in meta.thing_dataset_id_to_contiguous_id.items()}

        recs = []
        for p, nm in zip(rel, (GEN["name"], GEN2["name"])):
            recs += ref_ds.load_omni3d_json(p, "datasets", nm, fs_run, filter_empty=True)
        case["records"] = tolist(recs)
        case["repeat_factors"] = ref_build.repeat_factors_from_category_frequency(recs, 0.4).tolist()

        # mapper arithmetic on every record: resize (fixed) + flip for odd image ids
        mapped = []
        unknown = {len(case["thing_classes"])}
        for r in recs:
            h, w = r["height"], r["width"]
            tf = [d2data.ResizeTransform(h, w, int(h * 0.75), int(w * 0.75))]
            tf.append(d2data.HFlipTransform(int(w * 0.75)) if r["image_id"] % 2 else d2data.NoOpTransform())
            tl = d2data.TransformList(tf)
            K = np.array(r["K"])
            import copy
            annos = [ref_dm.transform_instance_annotations(copy.deepcopy(o), tl, K=K) for o in r["annotations"]]
            inst = ref_dm.annotations_to_instances(annos, (int(h * 0.75), int(w * 0.75)), unknown)
            mapped.append({"image_id": r["image_id"], "gt_classes": inst.gt_classes.tolist(),
                           "gt_boxes": inst.gt_boxes.tensor.tolist(), "gt_boxes3D": inst.gt_boxes3D.tolist(),
                           "gt_poses": inst.gt_poses.tolist(), "gt_keypoints": inst.gt_keypoints.tensor.tolist(),
                           "gt_unknown_category_mask": inst.gt_unknown_category_mask.tolist()})
        case["mapped"] = mapped

        cfg = syn.make_cfg(overrides=["DATASETS.MODAL_2D_BOXES", fs["modal_2D_boxes"], "DATASETS.TRUNC_2D_BOXES",
                                      fs["trunc_2D_boxes"]])
        case["priors_bins1"] = tolist(ref_math.compute_priors(cfg, omni, n_bins=1))
        case["priors_bins3"] = tolist(ref_math.compute_priors(cfg, omni, n_bins=3))
        out["cases"][sname] = case

    out["approx_eval_resolution"] = [[h, w, a, b, list(ref_math.approx_eval_resolution(h, w, a, b))]
                                     for h, w, a, b in ((480, 640, 512, 4096), (370, 1224, 512, 1000), (1920, 1080, 512, 4096))]
    with open(os.path.join(HERE, "data_path.json"), "w") as f:
        json.dump(out, f)
    print("wrote", os.path.join(HERE, "data_path.json"), os.path.getsize(os.path.join(HERE, "data_path.json")), "bytes")


if __name__ == "__main__":
    main()
