"""Omni3D data path against tests/golden/data_path.json, which holds the outputs of the
reference's own functions on the same seeded synthetic dataset (tests/golden/make_golden_data.py), plus host-logic
tests of the d2lite stand-ins (transforms, samplers, loaders)."""
import copy
import importlib
import json
import math
import os

import numpy as np
import pytest
import torch

syn = importlib.import_module("3dod_amd.synthetic")
data = importlib.import_module("3dod_amd.cubercnn.data")
D = importlib.import_module("3dod_amd.d2lite.data")
util = importlib.import_module("3dod_amd.cubercnn.util")

GOLD = os.path.join(os.path.dirname(__file__), "golden", "data_path.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def dataset_root(tmp_path_factory, gold):
    work = tmp_path_factory.mktemp("omni")
    root = work / "datasets"
    root.mkdir()
    paths = [syn.make_omni3d_dataset(str(root), **g) for g in gold["gen"]]
    return str(work), [os.path.relpath(p, str(work)) for p in paths]


def close(a, b, tol=1e-9):
    """nested lists / dicts / numbers, NaN == NaN"""
    if isinstance(b, dict):
        assert isinstance(a, dict) and set(map(str, a)) == set(b), (a, b)
        return all(close({str(k): v for k, v in a.items()}[k], b[k], tol) for k in b)
    if isinstance(b, (list, tuple)):
        assert len(a) == len(b), (len(a), len(b))
        return all(close(x, y, tol) for x, y in zip(a, b))
    if isinstance(b, float) or isinstance(a, float):
        if b is None or a is None:
            return a == b
        if math.isnan(b):
            return math.isnan(a)
        assert abs(a - b) <= tol * max(1.0, abs(b)), (a, b)
        return True
    assert a == b, (a, b)
    return True


def _plain(x):
    if isinstance(x, torch.Tensor) or isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.floating, np.integer, np.bool_)):
        return x.item()
    if isinstance(x, dict):
        return {str(k): _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


@pytest.mark.parametrize("sname", ["default", "modal_nearonly", "all_categories"])
def test_against_reference_outputs(gold, dataset_root, sname, monkeypatch, tmp_path):
    work, rel = dataset_root
    monkeypatch.chdir(work)
    case, fs = gold["cases"][sname], copy.deepcopy(gold["settings"][sname])

    # 1. is_ignore on every raw annotation
    raw = [json.load(open(p)) for p in rel]
    heights = {im["id"]: im["height"] for r in raw for im in r["images"]}
    got = {str(a["id"]): data.is_ignore(a, fs, heights[a["image_id"]]) for r in raw for a in r["annotations"]}
    assert got == case["is_ignore"]
    assert any(got.values()) and not all(got.values())

    # 2. the filtered COCO-style index
    omni = data.Omni3D(rel, filter_settings=fs)
    assert fs["category_names"] == case["omni3d"]["category_names_after"]
    assert [c["name"] for c in omni.dataset["categories"]] == case["omni3d"]["categories"]
    kept = [{k: a[k] for k in ("id", "area", "ignore", "bbox", "depth")} for a in omni.dataset["annotations"]]
    assert close(kept, case["omni3d"]["kept"])
    assert [i["known_category_ids"] for i in omni.dataset["info"]] == case["omni3d"]["known_category_ids"]

    # 3. model metadata, written once and re-read
    D.MetadataCatalog.pop("omni3d_model", None)
    data.register_and_store_model_metadata(omni, str(tmp_path), fs)
    meta = D.MetadataCatalog.get("omni3d_model")
    assert list(meta.thing_classes) == case["thing_classes"]
    assert {str(k): v for k, v in meta.thing_dataset_id_to_contiguous_id.items()} == case["id_map"]
    D.MetadataCatalog.pop("omni3d_model", None)
    data.register_and_store_model_metadata(None, str(tmp_path), None)          # second run: from category_meta.json
    meta = D.MetadataCatalog.get("omni3d_model")
    assert {str(k): v for k, v in meta.thing_dataset_id_to_contiguous_id.items()} == case["id_map"]
    assert all(isinstance(k, int) for k in meta.thing_dataset_id_to_contiguous_id)

    # 4. dataset dicts
    recs = []
    for p, g in zip(rel, gold["gen"]):
        recs += data.load_omni3d_json(p, "datasets", g["name"], fs, filter_empty=True)
    assert close(_plain(recs), case["records"])
    assert any("ground_image_path" in r for r in recs) and any("ground_image_path" not in r for r in recs)

    # 5. repeat factors
    rf = data.repeat_factors_from_category_frequency(recs, 0.4)
    assert close(rf.tolist(), case["repeat_factors"], 1e-6)

    # 6. mapper arithmetic: resize + flip (odd image ids), pose mirroring, gt_boxes3D packing
    unknown = {len(case["thing_classes"])}
    for r, exp in zip(recs, case["mapped"]):
        h, w = r["height"], r["width"]
        tf = [D.ResizeTransform(h, w, int(h * 0.75), int(w * 0.75)),
              D.HFlipTransform(int(w * 0.75)) if r["image_id"] % 2 else D.NoOpTransform()]
        annos = [data.transform_instance_annotations(copy.deepcopy(o), D.TransformList(tf), K=np.array(r["K"]))
                 for o in r["annotations"]]
        inst = data.annotations_to_instances(annos, (int(h * 0.75), int(w * 0.75)), unknown)
        assert r["image_id"] == exp["image_id"]
        assert inst.gt_classes.tolist() == exp["gt_classes"]
        for name, t in (("gt_boxes", inst.gt_boxes.tensor), ("gt_boxes3D", inst.gt_boxes3D), ("gt_poses", inst.gt_poses),
                        ("gt_keypoints", inst.gt_keypoints.tensor)):
            assert close(t.tolist(), exp[name], 1e-6), name
        assert inst.gt_unknown_category_mask.tolist() == exp["gt_unknown_category_mask"]

    # 7. priors
    cfg = syn.make_cfg(overrides=["DATASETS.MODAL_2D_BOXES", fs["modal_2D_boxes"], "DATASETS.TRUNC_2D_BOXES",
                                  fs["trunc_2D_boxes"]])
    for nb in (1, 3):
        pri = util.compute_priors(cfg, omni, n_bins=nb, category_names=case["thing_classes"])
        assert close(_plain(pri), case[f"priors_bins{nb}"], 2e-6), nb


def test_approx_eval_resolution(gold):
    for h, w, a, b, exp in gold["approx_eval_resolution"]:
        assert close(list(util.approx_eval_resolution(h, w, a, b)), exp)


def test_boxmode_and_transforms():
    assert D.BoxMode.convert([10, 20, 30, 60], D.BoxMode.XYXY_ABS, D.BoxMode.XYWH_ABS) == [10.0, 20.0, 20.0, 40.0]
    assert D.BoxMode.convert((10, 20, 20, 40), D.BoxMode.XYWH_ABS, D.BoxMode.XYXY_ABS) == (10.0, 20.0, 30.0, 60.0)
    arr = np.array([[1., 2., 3., 4.], [0., 0., 5., 6.]])
    assert np.array_equal(D.BoxMode.convert(arr, D.BoxMode.XYWH_ABS, D.BoxMode.XYXY_ABS), [[1, 2, 4, 6], [0, 0, 5, 6]])
    assert np.array_equal(arr, [[1., 2., 3., 4.], [0., 0., 5., 6.]])          # input untouched

    img = np.arange(4 * 6 * 3, dtype=np.uint8).reshape(4, 6, 3)
    fl = D.HFlipTransform(6)
    assert np.array_equal(fl.apply_image(img)[:, 0], img[:, 5])
    assert np.allclose(fl.apply_box(np.array([[1., 1., 3., 2.]])), [[3, 1, 5, 2]])
    rs = D.ResizeTransform(4, 6, 8, 9)
    assert rs.apply_image(img).shape == (8, 9, 3)
    assert np.allclose(rs.apply_coords(np.array([[2., 2.]])), [[3., 4.]])
    assert D.ResizeShortestEdge.get_output_shape(480, 640, 512, 4096) == (512, 683)
    assert D.ResizeShortestEdge.get_output_shape(370, 1224, 512, 1000) == (302, 1000)
    np.random.seed(0)
    aug = D.AugmentationList([D.ResizeShortestEdge((8, 12), 100, "choice"), D.RandomFlip(prob=1.0)])
    inp = D.AugInput(img)
    tl = aug(inp)
    assert isinstance(tl[0], D.ResizeTransform) and isinstance(tl[1], D.HFlipTransform)
    assert inp.image.shape[0] in (8, 12) and tl[1].width == inp.image.shape[1]


def test_samplers_shard_and_repeat():
    # every rank sees the same permutation stream, interleaved
    streams = [list(__import__("itertools").islice(iter(D.TrainingSampler(10, seed=3, rank=r, world_size=2)), 10))
               for r in range(2)]
    merged = [streams[i % 2][i // 2] for i in range(20)]
    assert sorted(merged[:10]) == list(range(10)) and sorted(merged[10:]) == list(range(10))
    assert merged[:10] == torch.randperm(10, generator=torch.Generator().manual_seed(3)).tolist()
    # inference: contiguous, complete, remainder to the first ranks
    shards = [list(D.InferenceSampler(11, rank=r, world_size=4)) for r in range(4)]
    assert [len(s) for s in shards] == [3, 3, 3, 2] and sum(shards, []) == list(range(11))
    # repeat factors: integer part always, fractional part stochastically
    rf = torch.tensor([1.0, 2.5, 1.0, 3.0])
    s = D.RepeatFactorTrainingSampler(rf, seed=1, rank=0, world_size=1)
    g = torch.Generator().manual_seed(1)
    epoch = s._get_epoch_indices(g).tolist()
    assert epoch.count(0) == 1 and epoch.count(3) == 3 and epoch.count(1) in (2, 3)
    counts = np.zeros(4)
    it = iter(s)
    for _ in range(7500):
        counts[next(it)] += 1
    assert abs(counts[1] / counts[0] - 2.5) < 0.15 and abs(counts[3] / counts[0] - 3.0) < 0.15


def _registered(work, rel, gold, fs):
    for n in list(D.DatasetCatalog):
        D.DatasetCatalog.remove(n)
    omni = data.Omni3D(rel, filter_settings=fs)
    D.MetadataCatalog.pop("omni3d_model", None)
    data.register_and_store_model_metadata(omni, work, fs)
    for g in gold["gen"]:
        data.simple_register(g["name"], fs, filter_empty=True)
    return omni


def test_loader_end_to_end(gold, dataset_root, monkeypatch):
    """json -> catalog -> mapper (real image decode, resize, flip, depth / ground maps) -> per-rank batches"""
    work, rel = dataset_root
    monkeypatch.chdir(work)
    fs = copy.deepcopy(gold["settings"]["default"])
    omni = _registered(work, rel, gold, fs)
    names = [g["name"] for g in gold["gen"]]
    cfg = syn.make_cfg(overrides=["DATASETS.TRAIN", tuple(names), "SOLVER.IMS_PER_BATCH", 4, "DATALOADER.NUM_WORKERS", 0,
                                  "INPUT.MIN_SIZE_TRAIN", (256, 320), "INPUT.MAX_SIZE_TRAIN", 512, "SEED", 5])
    meta = D.MetadataCatalog.get("omni3d_model")
    unknown, id_to_src = data.build.dataset_id_maps(omni, cfg.MODEL.ROI_HEADS.NUM_CLASSES,
                                                    meta.thing_dataset_id_to_contiguous_id)
    assert set(id_to_src.values()) == {"synthetic", "synthetic_b"}
    assert cfg.MODEL.ROI_HEADS.NUM_CLASSES in unknown[90]

    mapper = data.DatasetMapper3D(cfg, is_train=True)
    mapper.dataset_id_to_unknown_cats = unknown
    np.random.seed(0)
    per_rank = []
    for rank in range(2):
        loader = data.build_detection_train_loader(cfg, mapper=mapper, dataset_id_to_src=id_to_src, rank=rank, world_size=2)
        it = iter(loader)
        batches = [next(it) for _ in range(3)]
        per_rank.append(batches)
        for b in batches:
            assert len(b) == 2                                       # global 4 / 2 ranks
            assert len({d["width"] > d["height"] for d in b}) == 1   # aspect-ratio grouping
            for d in b:
                img = d["image"]
                assert img.dtype == torch.uint8 and img.shape[0] == 3 and min(img.shape[1:]) in (256, 320)
                assert d["depth_map"].shape == img.shape[1:]
                assert d["ground_map"] is None or (d["ground_map"].shape == img.shape[1:] and d["ground_map"].dtype == torch.bool)
                inst = d["instances"]
                assert inst.image_size == tuple(img.shape[1:]) and len(inst) > 0
                assert inst.gt_boxes3D.shape[1] == 9 and inst.gt_poses.shape[1:] == (3, 3)
                assert (inst.gt_boxes.tensor[:, 2:] > inst.gt_boxes.tensor[:, :2]).all()
                R = inst.gt_poses
                assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand_as(R), atol=1e-5)
    ids = [[d["image_id"] for b in batches for d in b] for batches in per_rank]
    assert ids[0] != ids[1]

    # the decoded image really is the file's pixels: BGR of the RGB png, resized
    rec = D.DatasetCatalog.get(names[0])[0]
    test_mapper = data.DatasetMapper3D(cfg, is_train=False)
    out = test_mapper(rec)
    from PIL import Image
    rgb = np.asarray(Image.open(rec["file_name"]).convert("RGB"))
    h, w = rgb.shape[:2]
    nh, nw = D.ResizeShortestEdge.get_output_shape(h, w, cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST)
    exp = np.asarray(Image.fromarray(rgb[:, :, ::-1].copy()).resize((nw, nh), Image.BILINEAR))
    assert np.array_equal(out["image"].numpy(), exp.transpose(2, 0, 1))
    assert "instances" not in out and "annotations" in out

    # test loader: rank shards cover the dataset once, in order
    seen = []
    for rank in range(2):
        tl = data.build_detection_test_loader(cfg, names[0], mapper=test_mapper, batch_size=2, rank=rank, world_size=2,
                                              num_workers=0)
        seen += [d["image_id"] for b in tl for d in b]
    assert seen == [r["image_id"] for r in data.get_detection_dataset_dicts(names[0], filter_empty=False)]


def test_flip_is_an_involution_on_annotations():
    K = np.array([[500., 0, 320], [0, 500., 240], [0, 0, 1]])
    yaw = 0.7
    R = [[math.cos(yaw), 0, math.sin(yaw)], [0, 1, 0], [-math.sin(yaw), 0, math.cos(yaw)]]
    corners = (np.random.default_rng(0).uniform(-0.5, 0.5, (8, 3)) + np.array([0.4, 0.1, 5.0])).tolist()
    obj = {"bbox": [300., 200., 80., 60.], "bbox_mode": D.BoxMode.XYWH_ABS, "center_cam": [0.4, 0.1, 5.0],
           "bbox3D_cam": corners, "dimensions": [1., 1., 1.], "pose": R, "ignore": False, "category_id": 0}
    once = data.transform_instance_annotations(copy.deepcopy(obj), [D.HFlipTransform(640)], K=K)
    assert np.allclose(once["bbox"], [640 - 380, 200, 640 - 300, 260])
    assert np.allclose(once["center_cam_proj"][0], 640 - (500 * 0.4 / 5 + 320))
    # mirrored pose: yaw -> -yaw for a rotation about y (up to the sign convention M1 R M2)
    P = np.array(once["pose"])
    assert np.allclose(P @ P.T, np.eye(3)) and np.isclose(np.linalg.det(P), 1.0)
    # a second flip mirrors the pose back (M1 M1 = M2 M2 = I) and restores box and centre
    twice = data.transform_instance_annotations(copy.deepcopy(obj), [D.HFlipTransform(640), D.HFlipTransform(640)], K=K)
    assert np.allclose(twice["pose"], R) and np.allclose(twice["bbox"], [300, 200, 380, 260])
    assert np.allclose(twice["center_cam_proj"][:2], [500 * 0.4 / 5 + 320, 500 * 0.1 / 5 + 240])
    assert np.allclose(np.array(once["pose"]), np.diag([1., -1, -1]) @ np.array(R) @ np.diag([-1., -1, 1]))


def test_builtin_categories():
    assert len(data.get_omni3d_categories("omni3d")) == 50
    assert data.get_omni3d_categories("omni3d_in") | data.get_omni3d_categories("omni3d_out") \
        <= data.get_omni3d_categories("omni3d")
    assert data.get_omni3d_categories("KITTI_val") == {"pedestrian", "car", "cyclist", "van", "truck"}
    assert "toilet" in data.get_omni3d_categories("Hypersim_val") and "toilet" not in data.get_omni3d_categories("Hypersim_test")
    with pytest.raises(ValueError):
        data.get_omni3d_categories("nope")


def test_evaluator_over_datasets(gold, dataset_root, monkeypatch, tmp_path):
    """json ground truth -> Omni3DEvaluator / Omni3DEvaluationHelper: detections equal to the valid ground truth score
    AP2D = AP3D = 100 per dataset and pooled; class ids travel contiguous -> Omni3D ids; classes a dataset does not
    annotate are dropped; a shifted copy at lower score does not hurt, a confident false positive does."""
    from oracle import iou3d as oiou
    ev_mod = importlib.import_module("3dod_amd.cubercnn.evaluation")
    work, rel = dataset_root
    monkeypatch.chdir(work)
    fs = copy.deepcopy(gold["settings"]["default"])
    omni = _registered(work, rel, gold, fs)
    names = [g["name"] for g in gold["gen"]]
    meta = D.MetadataCatalog.get("omni3d_model")
    id_map = meta.thing_dataset_id_to_contiguous_id

    def iou3d(d, g):
        return oiou.box3d_overlap(np.asarray(d, np.float64), np.asarray(g, np.float64))[1]

    def predictions(name, extra_fp=False, oov=False):
        api = data.Omni3D([os.path.join("datasets", "Omni3D", name + ".json")], copy.deepcopy(fs))
        by_img = {}
        for a in api.dataset["annotations"]:
            inst = {"image_id": a["image_id"], "category_id": id_map[a["category_id"]], "bbox": list(a["bbox"]),
                    "score": 0.9, "depth": a["depth"], "bbox3D": a["bbox3D"]}
            by_img.setdefault(a["image_id"], []).append(inst)
            shifted = copy.deepcopy(inst)
            shifted["score"] = 0.2
            shifted["bbox"][0] += 3.0
            by_img[a["image_id"]].append(shifted)
            if extra_fp:
                fp = copy.deepcopy(inst)
                fp["score"] = 0.99
                fp["bbox"] = [5.0, 5.0, 12.0, 9.0]
                fp["bbox3D"] = (np.asarray(a["bbox3D"]) + 40.0).tolist()
                by_img[a["image_id"]].append(fp)
            if oov:                      # a class of the model that this dataset does not annotate
                o = copy.deepcopy(inst)
                o["category_id"], o["score"] = id_map[2], 1.0          # 'bed' (id 2) is not in Synth_b_train
                by_img[a["image_id"]].append(o)
        return [{"image_id": i, "K": api.imgs[i]["K"], "width": api.imgs[i]["width"], "height": api.imgs[i]["height"],
                 "instances": v} for i, v in by_img.items()]

    helper = ev_mod.Omni3DEvaluationHelper(names, fs, str(tmp_path), iter_label="t", iou3d_fn=iou3d)
    helper.add_predictions(names[0], predictions(names[0]))
    helper.add_predictions(names[1], predictions(names[1], oov=True))
    for n in names:
        res = helper.evaluate(n)
        assert res["bbox_2D"]["AP"] == pytest.approx(100.0) and res["bbox_3D"]["AP"] == pytest.approx(100.0), res
        assert os.path.exists(os.path.join(str(tmp_path), n, "omni_instances_results.json"))
    assert "AP-bed" in helper.results[names[0]]["bbox_2D"] and "AP-bed" not in helper.results[names[1]]["bbox_2D"]
    with open(os.path.join(str(tmp_path), names[1], "omni_instances_results.json")) as f:
        saved = json.load(f)
    assert saved and all(r["category_id"] in (5, 8) for r in saved)          # Omni3D ids of car / chair; 'bed' dropped
    analysis, omni_tab = helper.summarize_all()
    assert analysis["<Concat>"]["AP2D"] == pytest.approx(100.0) and analysis["<Concat>"]["AP3D"] == pytest.approx(100.0)
    assert math.isnan(omni_tab["Omni3D"]["AP3D"])                             # not all 50 categories present
    assert analysis[names[0]]["AP3D@25"] == pytest.approx(100.0)

    worse = ev_mod.Omni3DEvaluationHelper(names[:1], fs, str(tmp_path / "fp"), iou3d_fn=iou3d)
    worse.add_predictions(names[0], predictions(names[0], extra_fp=True))
    res = worse.evaluate(names[0])
    assert 20.0 < res["bbox_2D"]["AP"] < 80.0 and 20.0 < res["bbox_3D"]["AP"] < 80.0, res


def test_instances_to_coco_json():
    ev_mod = importlib.import_module("3dod_amd.cubercnn.evaluation")
    d2 = importlib.import_module("3dod_amd.d2lite")
    inst = d2.Instances((100, 200))
    inst.pred_boxes = d2.Boxes(torch.tensor([[10., 20., 50., 80.], [0., 0., 5., 5.]]))
    inst.scores = torch.tensor([0.9, 0.1])
    inst.pred_classes = torch.tensor([3, 1])
    out = ev_mod.instances_to_coco_json(inst, 77)
    assert out[0]["bbox"] == [10.0, 20.0, 40.0, 60.0] and out[0]["image_id"] == 77 and out[1]["category_id"] == 1
    assert out[0]["depth"] == 1.0 and np.array(out[0]["bbox3D"]).shape == (8, 3)          # placeholders without a 3D head
    corners = torch.arange(48, dtype=torch.float32).reshape(2, 8, 3)
    inst.pred_bbox3D, inst.pred_center_cam, inst.pred_center_2D = corners, torch.ones(2, 3), torch.ones(2, 2)
    inst.pred_dimensions, inst.pred_pose = torch.ones(2, 3), torch.eye(3).expand(2, 3, 3)
    out = ev_mod.instances_to_coco_json(inst, 78)
    assert out[1]["depth"] == pytest.approx(float(corners[1, :, 2].mean()))
    assert ev_mod.instances_to_coco_json(inst[torch.zeros(2, dtype=torch.bool)], 1) == []


def test_loader_workers_draw_different_augmentations(gold, dataset_root, monkeypatch):
    """two loader workers must not replay the same numpy random stream (flip / scale choices)"""
    work, rel = dataset_root
    monkeypatch.chdir(work)
    fs = copy.deepcopy(gold["settings"]["default"])
    omni = _registered(work, rel, gold, fs)
    names = [g["name"] for g in gold["gen"]]
    cfg = syn.make_cfg(overrides=["DATASETS.TRAIN", tuple(names), "SOLVER.IMS_PER_BATCH", 2, "DATALOADER.NUM_WORKERS", 2,
                                  "DATALOADER.ASPECT_RATIO_GROUPING", False, "INPUT.MIN_SIZE_TRAIN", tuple(range(128, 257, 8)),
                                  "INPUT.MAX_SIZE_TRAIN", 512])
    meta = D.MetadataCatalog.get("omni3d_model")
    unknown, id_to_src = data.build.dataset_id_maps(omni, cfg.MODEL.ROI_HEADS.NUM_CLASSES, meta.thing_dataset_id_to_contiguous_id)
    mapper = data.DatasetMapper3D(cfg, is_train=True)
    with pytest.raises(RuntimeError, match="dataset_id_to_unknown_cats"):
        mapper(D.DatasetCatalog.get(names[0])[0])
    mapper.dataset_id_to_unknown_cats = unknown
    torch.manual_seed(0)
    it = iter(data.build_detection_train_loader(cfg, mapper=mapper, dataset_id_to_src=id_to_src, rank=0, world_size=1))
    sizes = [min(d["image"].shape[1:]) for _ in range(6) for d in next(it)]
    # worker 0 produces batches 0, 2, 4 and worker 1 batches 1, 3, 5: with cloned RNG states the two would pick the same
    # sequence of short edges
    w0, w1 = sizes[0:2] + sizes[4:6] + sizes[8:10], sizes[2:4] + sizes[6:8] + sizes[10:12]
    assert w0 != w1, (w0, w1)
