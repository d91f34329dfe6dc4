"""Exploration behind tests/test_gpu_precision_parity.py: the same training run (same seed, same batches) in the reference's
float32 arithmetic and in the bf16 fast mode -- per-step loss trajectories, then AP2D / AP3D of both models on the
memorised synthetic set through Omni3DEvaluationHelper, and the fp32-trained weights evaluated in both modes.

    python scripts/precision_parity.py [steps] [n_images]
"""
import copy
import importlib
import json
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
syn = importlib.import_module("3dod_amd.synthetic")
data = importlib.import_module("3dod_amd.cubercnn.data")
D = importlib.import_module("3dod_amd.d2lite.data")
d2 = importlib.import_module("3dod_amd.d2lite")
util = importlib.import_module("3dod_amd.cubercnn.util")
modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
solver = importlib.import_module("3dod_amd.cubercnn.solver")
ev_mod = importlib.import_module("3dod_amd.cubercnn.evaluation")
ops = importlib.import_module("3dod_amd.hipops")


def setup(tmp, n_images, seed=5):
    dev = torch.device("cuda:0")
    root = os.path.join(tmp, "datasets")
    os.makedirs(root, exist_ok=True)
    syn.make_omni3d_dataset(root, name="Synth_mem", n_images=n_images, seed=seed, sizes=((512, 512),))
    os.chdir(tmp)
    for n in list(D.DatasetCatalog):
        D.DatasetCatalog.remove(n)
    for n in ("omni3d_model", "Synth_mem"):
        D.MetadataCatalog.pop(n, None)
    cats = ["bed", "car", "chair", "sofa", "table", "truck"]
    cfg = syn.make_cfg(overrides=[
        "MODEL.DEVICE", str(dev), "DATASETS.TRAIN", ("Synth_mem",), "DATASETS.TEST", ("Synth_mem",),
        "DATASETS.CATEGORY_NAMES", cats, "MODEL.ROI_HEADS.NUM_CLASSES", len(cats), "SOLVER.IMS_PER_BATCH", 4,
        "DATALOADER.NUM_WORKERS", 0, "INPUT.MIN_SIZE_TRAIN", (512,), "INPUT.MAX_SIZE_TRAIN", 512, "INPUT.MIN_SIZE_TEST", 512,
        "INPUT.MAX_SIZE_TEST", 512, "INPUT.RANDOM_FLIP", "none", "SOLVER.BASE_LR", 0.0025, "VIS_PERIOD", 0, "log", False,
        "SEED", 1])
    fs = data.get_filter_settings_from_cfg(cfg)
    api = data.Omni3D([os.path.join("datasets", "Omni3D", "Synth_mem.json")], copy.deepcopy(fs))
    data.register_and_store_model_metadata(api, tmp, fs)
    data.simple_register("Synth_mem", fs, filter_empty=True)
    meta = D.MetadataCatalog.get("omni3d_model")
    unknown, id_to_src = data.build.dataset_id_maps(api, len(cats), meta.thing_dataset_id_to_contiguous_id)
    priors = util.compute_priors(cfg, api)
    mapper = data.DatasetMapper3D(cfg, is_train=True)
    mapper.dataset_id_to_unknown_cats = unknown
    # a fixed list of training batches (the memorised set), resident on the device
    loader = iter(data.build_detection_train_loader(cfg, mapper=mapper, dataset_id_to_src=id_to_src, rank=0, world_size=1,
                                                    num_workers=0))
    np.random.seed(0)
    torch.manual_seed(0)
    batches = []
    for _ in range(max(1, n_images // 4)):
        b = next(loader)
        for d in b:
            d["image"] = d["image"].to(dev)
            d["instances"] = d["instances"].to(dev)
        batches.append(b)
    return dev, cfg, fs, priors, batches


def train(cfg, priors, batches, prec, steps, seed=0):
    ops.set_precision(prec)
    torch.manual_seed(seed)
    model = modeling.build_model(cfg, priors=priors).train()
    opt = solver.build_optimizer(cfg, model)
    step = solver.TrainStep(cfg, model, opt, world_size=1)
    sched = solver.WarmupMultiStepLR(opt, [], 0.1, 0.01, 50, "linear", None)
    torch.manual_seed(seed + 1)            # the sampling draws of both runs come from the same stream
    traj = []
    with d2.EventStorage(0):
        for i in range(steps):
            step(batches[i % len(batches)])
            sched.step()
            traj.append(step.report()["total_loss"])
    return model, traj


def evaluate(cfg, fs, model, prec, tag, tmp):
    ops.set_precision(prec)
    model.eval()
    loader = data.build_detection_test_loader(cfg, "Synth_mem", batch_size=4, rank=0, world_size=1, num_workers=0)
    out = ev_mod.inference_on_dataset(model, loader)
    helper = ev_mod.Omni3DEvaluationHelper(["Synth_mem"], fs, os.path.join(tmp, "eval_" + tag))
    helper.add_predictions("Synth_mem", out)
    res = helper.evaluate("Synth_mem")
    return res["bbox_2D"]["AP"], res["bbox_3D"]["AP"], sum(len(p["instances"]) for p in out)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    n_images = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    tmp = tempfile.mkdtemp(prefix="cr_parity_")
    dev, cfg, fs, priors, batches = setup(tmp, n_images)
    res = {}
    models = {}
    for prec in ("fp32", "bf16"):
        models[prec], traj = train(cfg, priors, batches, prec, steps)
        res["traj_" + prec] = traj
        print(prec, "losses:", " ".join(f"{v:.3f}" for v in traj[:5]), "...", " ".join(f"{v:.3f}" for v in traj[-5:]), flush=True)
    a, b = np.array(res["traj_fp32"]), np.array(res["traj_bf16"])
    rel = np.abs(a - b) / np.abs(a)
    for k in (1, 5, 10, 20, 50, 100, steps):
        print(f"max rel loss diff over first {k:4d} steps: {rel[:k].max():.4f}   mean {rel[:k].mean():.4f}")
    w = 20
    sm = lambda v: np.convolve(v, np.ones(w) / w, mode="valid")
    print(f"smoothed ({w}) max rel diff: {np.abs(sm(a) - sm(b)).max() / sm(a).min():.4f}; final smoothed {sm(a)[-1]:.3f} vs {sm(b)[-1]:.3f}")
    for name, m, prec in (("fp32-trained @fp32", models["fp32"], "fp32"), ("fp32-trained @bf16", models["fp32"], "bf16"),
                          ("bf16-trained @bf16", models["bf16"], "bf16"), ("bf16-trained @fp32", models["bf16"], "fp32")):
        ap2, ap3, n = evaluate(cfg, fs, m, prec, name.replace(" ", "_").replace("@", ""), tmp)
        print(f"{name}: AP2D {ap2:.4f}  AP3D {ap3:.4f}  detections {n}", flush=True)
        res[name] = (ap2, ap3, n)
    json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "precision_parity.json"), "w"))


if __name__ == "__main__":
    main()
