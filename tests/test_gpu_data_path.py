"""Data path feeding the device: Omni3D json -> loader -> DevicePrefetcher (pinned staging + side-stream H2D) ->
Cube R-CNN train steps built with priors computed from the same annotations; then the test loader -> inference."""
import copy
import importlib
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

syn = importlib.import_module("3dod_amd.synthetic")
data = importlib.import_module("3dod_amd.cubercnn.data")
D = importlib.import_module("3dod_amd.d2lite.data")
d2 = importlib.import_module("3dod_amd.d2lite")
util = importlib.import_module("3dod_amd.cubercnn.util")
modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
solver = importlib.import_module("3dod_amd.cubercnn.solver")


def test_loader_prefetch_train_and_infer(tmp_path, monkeypatch):
    dev = torch.device("cuda:0")
    root = tmp_path / "datasets"
    root.mkdir()
    syn.make_omni3d_dataset(str(root), name="Synth_train", n_images=16, seed=3)
    monkeypatch.chdir(tmp_path)
    for n in list(D.DatasetCatalog):
        D.DatasetCatalog.remove(n)
    D.MetadataCatalog.pop("omni3d_model", None)
    cats = ["bed", "car", "chair", "sofa", "table", "truck"]
    cfg = syn.make_cfg(overrides=[
        "MODEL.DEVICE", str(dev), "DATASETS.TRAIN", ("Synth_train",), "DATASETS.CATEGORY_NAMES", cats,
        "MODEL.ROI_HEADS.NUM_CLASSES", len(cats), "SOLVER.IMS_PER_BATCH", 2, "DATALOADER.NUM_WORKERS", 2,
        "INPUT.MIN_SIZE_TRAIN", (256, 288), "INPUT.MAX_SIZE_TRAIN", 512, "INPUT.MIN_SIZE_TEST", 256,
        "INPUT.MAX_SIZE_TEST", 512, "SOLVER.BASE_LR", 0.00125, "VIS_PERIOD", 0, "log", False, "SEED", 1])
    fs = data.get_filter_settings_from_cfg(cfg)
    rel = [os.path.join("datasets", "Omni3D", "Synth_train.json")]
    omni = data.Omni3D(rel, filter_settings=fs)
    data.register_and_store_model_metadata(omni, str(tmp_path), fs)
    data.simple_register("Synth_train", fs, filter_empty=True)
    meta = D.MetadataCatalog.get("omni3d_model")
    assert meta.thing_classes == cats
    unknown, id_to_src = data.build.dataset_id_maps(omni, len(cats), meta.thing_dataset_id_to_contiguous_id)
    priors = util.compute_priors(cfg, omni)
    assert len(priors["priors_dims_per_cat"]) == len(cats)

    mapper = data.DatasetMapper3D(cfg, is_train=True)
    mapper.dataset_id_to_unknown_cats = unknown
    loader = data.build_detection_train_loader(cfg, mapper=mapper, dataset_id_to_src=id_to_src, rank=0, world_size=1)

    # staged batches equal the host batches (same seeded stream read twice)
    host_it = iter(data.build_detection_train_loader(cfg, mapper=mapper, dataset_id_to_src=id_to_src, rank=0,
                                                     world_size=1, num_workers=0))
    np.random.seed(0)
    host = [next(host_it) for _ in range(2)]
    np.random.seed(0)
    pre = data.DevicePrefetcher(iter(data.build_detection_train_loader(
        cfg, mapper=mapper, dataset_id_to_src=id_to_src, rank=0, world_size=1, num_workers=0)), dev)
    for hb in host:
        db = next(pre)
        for h, d in zip(hb, db):
            assert d["image"].is_cuda and torch.equal(d["image"].cpu(), h["image"])
            assert torch.equal(d["depth_map"].cpu(), h["depth_map"])
            assert d["instances"].gt_boxes.tensor.is_cuda
            assert torch.equal(d["instances"].gt_boxes3D.cpu(), h["instances"].gt_boxes3D)
            assert torch.equal(d["instances"].gt_classes.cpu(), h["instances"].gt_classes)

    torch.manual_seed(0)
    model = modeling.build_model(cfg, priors=priors).train()
    opt = solver.build_optimizer(cfg, model)
    step = solver.TrainStep(cfg, model, opt, world_size=1)
    feed = data.DevicePrefetcher(loader, dev)
    with d2.EventStorage(0):
        for _ in range(3):
            step(next(feed))
        rep = step.report()
    assert rep["total_loss"] == rep["total_loss"] and abs(rep["total_loss"]) < 1e4, rep

    model.eval()
    tl = data.build_detection_test_loader(cfg, "Synth_train", batch_size=2, rank=0, world_size=1, num_workers=0)
    n = 0
    with torch.no_grad():
        for batch in tl:
            outs = model(batch)
            assert len(outs) == len(batch)
            for o, b in zip(outs, batch):
                inst = o["instances"]
                assert inst.image_size == (b["height"], b["width"])
                if len(inst):
                    assert inst.pred_bbox3D.shape[1:] == (8, 3) and int(inst.pred_classes.max()) < len(cats)
                n += 1
    assert n == len(D.DatasetCatalog.get("Synth_train"))


def test_evaluation_with_device_iou(tmp_path, monkeypatch):
    """ground-truth cuboids as detections through Omni3DEvaluationHelper with the exact-IoU3D kernel: AP3D = 100;
    then real (random-weight) detections through inference_on_dataset -> evaluator run end to end."""
    ev_mod = importlib.import_module("3dod_amd.cubercnn.evaluation")
    dev = torch.device("cuda:0")
    root = tmp_path / "datasets"
    root.mkdir()
    syn.make_omni3d_dataset(str(root), name="Synth_val", n_images=8, seed=11)
    monkeypatch.chdir(tmp_path)
    for n in list(D.DatasetCatalog):
        D.DatasetCatalog.remove(n)
    for n in ("omni3d_model", "Synth_val"):
        D.MetadataCatalog.pop(n, None)
    cats = ["bed", "car", "chair", "sofa", "table", "truck"]
    cfg = syn.make_cfg(overrides=["MODEL.DEVICE", str(dev), "DATASETS.TEST", ("Synth_val",), "DATASETS.CATEGORY_NAMES", cats,
                                  "MODEL.ROI_HEADS.NUM_CLASSES", len(cats), "INPUT.MIN_SIZE_TEST", 256,
                                  "INPUT.MAX_SIZE_TEST", 512, "VIS_PERIOD", 0, "log", False])
    fs = data.get_filter_settings_from_cfg(cfg)
    api = data.Omni3D([os.path.join("datasets", "Omni3D", "Synth_val.json")], copy.deepcopy(fs))
    data.register_and_store_model_metadata(api, str(tmp_path), fs)
    id_map = D.MetadataCatalog.get("omni3d_model").thing_dataset_id_to_contiguous_id

    by_img = {}
    for a in api.dataset["annotations"]:
        by_img.setdefault(a["image_id"], []).append(
            {"image_id": a["image_id"], "category_id": id_map[a["category_id"]], "bbox": list(a["bbox"]), "score": 0.8,
             "depth": a["depth"], "bbox3D": a["bbox3D"]})
    preds = [{"image_id": i, "K": api.imgs[i]["K"], "width": api.imgs[i]["width"], "height": api.imgs[i]["height"],
              "instances": v} for i, v in by_img.items()]
    helper = ev_mod.Omni3DEvaluationHelper(["Synth_val"], fs, str(tmp_path / "eval_gt"))
    helper.add_predictions("Synth_val", preds)
    res = helper.evaluate("Synth_val")
    assert res["bbox_3D"]["AP"] == pytest.approx(100.0) and res["bbox_2D"]["AP"] == pytest.approx(100.0), res

    torch.manual_seed(0)
    model = modeling.build_model(cfg, priors=util.compute_priors(cfg, api)).eval()
    loader = data.build_detection_test_loader(cfg, "Synth_val", batch_size=4, rank=0, world_size=1, num_workers=0)
    out = ev_mod.inference_on_dataset(model, loader)
    assert len(out) == len(D.DatasetCatalog.get("Synth_val")) and not model.training
    helper = ev_mod.Omni3DEvaluationHelper(["Synth_val"], fs, str(tmp_path / "eval_model"))
    helper.add_predictions("Synth_val", out)
    analysis, _ = helper.summarize_all()
    if any(p["instances"] for p in out):
        assert 0.0 <= analysis["<Concat>"]["AP2D"] <= 100.0 and 0.0 <= analysis["<Concat>"]["AP3D"] <= 100.0
