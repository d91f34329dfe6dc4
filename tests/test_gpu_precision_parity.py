"""Model-level parity of the two arithmetic modes and of the HIP path with the float32 CPU oracle.

north_star: "Outputs match the reference on identical inputs ... AP3D within 1e-3".  The reference trains and evaluates in
float32 (tools/train_net.py:184-330, no autocast), so:

  * the DEFAULT mode of this build is float32 (f32 MFMA) and the AP3D criterion is checked on it: the same weights give
    the same AP2D / AP3D (within 1e-3 AP points, the evaluator reports percent) through the HIP kernels and through the
    float32 CPU oracle (oracle/cpu_backend.py: the tensor-op formulation in ATen float32) on the memorised synthetic set,
    scored by Omni3DEvaluationHelper (N1) with the exact-IoU3D kernel;
  * the bf16 fast mode is opt-in and its deviation is MEASURED here, not claimed away: loss trajectories of the same run
    (same seed, same batches, same sampling stream) agree to 1 % over the first 5 steps, 3 % over the first 10 and 10 %
    over the first 20 (after that the two runs pick different RoI samples and decorrelate like two seeds -- or two
    float32 runs with their float atomics -- do); inference with the SAME weights moves
    AP3D by a few points on this 8-image set (one detection is ~1 AP point there) -- bounded at 5 points.  That is why the
    headline benchmark is measured in float32.
"""
import copy
import importlib
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
ev_mod = importlib.import_module("3dod_amd.cubercnn.evaluation")
ops = importlib.import_module("3dod_amd.hipops")
CATS = ["bed", "car", "chair", "sofa", "table", "truck"]


def _setup(tmp_path, monkeypatch, n_images=8):
    dev = torch.device("cuda:0")
    root = tmp_path / "datasets"
    root.mkdir()
    syn.make_omni3d_dataset(str(root), name="Synth_mem", n_images=n_images, seed=5, sizes=((512, 512),))
    monkeypatch.chdir(tmp_path)
    for n in list(D.DatasetCatalog):
        D.DatasetCatalog.remove(n)
    for n in ("omni3d_model", "Synth_mem"):
        D.MetadataCatalog.pop(n, None)
    over = ["DATASETS.TRAIN", ("Synth_mem",), "DATASETS.TEST", ("Synth_mem",), "DATASETS.CATEGORY_NAMES", CATS,
            "MODEL.ROI_HEADS.NUM_CLASSES", len(CATS), "SOLVER.IMS_PER_BATCH", 4, "DATALOADER.NUM_WORKERS", 0,
            "INPUT.MIN_SIZE_TRAIN", (512,), "INPUT.MAX_SIZE_TRAIN", 512, "INPUT.MIN_SIZE_TEST", 512, "INPUT.MAX_SIZE_TEST", 512,
            "INPUT.RANDOM_FLIP", "none", "SOLVER.BASE_LR", 0.0025, "VIS_PERIOD", 0, "log", False, "SEED", 1]
    cfg = syn.make_cfg(overrides=["MODEL.DEVICE", str(dev)] + over)
    fs = data.get_filter_settings_from_cfg(cfg)
    api = data.Omni3D([os.path.join("datasets", "Omni3D", "Synth_mem.json")], copy.deepcopy(fs))
    data.register_and_store_model_metadata(api, str(tmp_path), fs)
    data.simple_register("Synth_mem", fs, filter_empty=True)
    meta = D.MetadataCatalog.get("omni3d_model")
    unknown, id_to_src = data.build.dataset_id_maps(api, len(CATS), meta.thing_dataset_id_to_contiguous_id)
    priors = util.compute_priors(cfg, api)
    mapper = data.DatasetMapper3D(cfg, is_train=True)
    mapper.dataset_id_to_unknown_cats = unknown
    loader = iter(data.build_detection_train_loader(cfg, mapper=mapper, dataset_id_to_src=id_to_src, rank=0, world_size=1,
                                                    num_workers=0))
    np.random.seed(0)
    torch.manual_seed(0)
    batches = []
    for _ in range(n_images // 4):                        # the memorised set: fixed batches resident on the device
        b = next(loader)
        for d in b:
            d["image"], d["instances"] = d["image"].to(dev), d["instances"].to(dev)
        batches.append(b)
    return dev, cfg, over, fs, priors, batches


def _train(cfg, priors, batches, prec, steps):
    prev = ops.set_precision(prec)
    try:
        torch.manual_seed(0)
        model = modeling.build_model(cfg, priors=priors).train()
        opt = solver.build_optimizer(cfg, model)
        step = solver.TrainStep(cfg, model, opt, world_size=1)
        sched = solver.WarmupMultiStepLR(opt, [], 0.1, 0.01, 50, "linear", None)
        torch.manual_seed(1)                              # both runs draw their sampling variates from the same stream
        traj = []
        with d2.EventStorage(0):
            for i in range(steps):
                step(batches[i % len(batches)])
                sched.step()
                if i < 20 or i == steps - 1:
                    traj.append(step.report()["total_loss"])
        assert step.report()["iterations_explode"] == 0
    finally:
        ops.set_precision(prev)
    return model, traj


def _evaluate(cfg, fs, model, tag, tmp_path, batch_size=4):
    model.eval()
    loader = data.build_detection_test_loader(cfg, "Synth_mem", batch_size=batch_size, rank=0, world_size=1, num_workers=0)
    out = ev_mod.inference_on_dataset(model, loader)
    helper = ev_mod.Omni3DEvaluationHelper(["Synth_mem"], fs, str(tmp_path / ("eval_" + tag)))
    helper.add_predictions("Synth_mem", out)
    res = helper.evaluate("Synth_mem")
    return float(res["bbox_2D"]["AP"]), float(res["bbox_3D"]["AP"]), out


def _compare_detections(a, b, min_score=0.05):
    """prediction records of two runs over the same images: detections with score >= min_score in either run are paired by
    (image, class, 2D IoU > 0.9); returns ({field: worst relative deviation over the pairs}, pairs, detections considered)"""
    by_img = lambda recs: {r["image_id"]: [d for d in r["instances"] if d["score"] >= min_score] for r in recs}
    A, B = by_img(a), by_img(b)
    worst = {"score": 0.0, "bbox": 0.0, "center_cam": 0.0, "dimensions": 0.0, "bbox3D": 0.0}
    n_match = n_tot = 0

    def iou(p, q):
        ax0, ay0, aw, ah = p
        bx0, by0, bw, bh = q
        iw = max(0.0, min(ax0 + aw, bx0 + bw) - max(ax0, bx0))
        ih = max(0.0, min(ay0 + ah, by0 + bh) - max(ay0, by0))
        return iw * ih / max(aw * ah + bw * bh - iw * ih, 1e-12)
    for img in A:
        da, db = A[img], list(B.get(img, []))
        n_tot += max(len(da), len(db))
        for d in da:
            cand = [(iou(d["bbox"], e["bbox"]), j) for j, e in enumerate(db) if e["category_id"] == d["category_id"]]
            if not cand or max(cand)[0] <= 0.9:
                continue
            e = db.pop(max(cand)[1])
            n_match += 1
            worst["score"] = max(worst["score"], abs(d["score"] - e["score"]) / max(abs(e["score"]), 0.05))
            for k, scale in (("bbox", 512.0), ("center_cam", None), ("dimensions", None), ("bbox3D", None)):
                x, y = np.asarray(d[k], np.float64), np.asarray(e[k], np.float64)
                den = scale if scale is not None else max(float(np.abs(y).max()), 1e-6)
                worst[k] = max(worst[k], float(np.abs(x - y).max()) / den)
    return worst, n_match, n_tot


def test_loss_trajectory_bf16_mode_follows_fp32(tmp_path, monkeypatch):
    dev, cfg, over, fs, priors, batches = _setup(tmp_path, monkeypatch)
    _, a = _train(cfg, priors, batches, "fp32", 20)
    _, b = _train(cfg, priors, batches, "bf16", 20)
    a, b = np.array(a[:20]), np.array(b[:20])
    rel = np.abs(a - b) / np.abs(a)
    assert a[-1] < a[0], "the float32 run must be learning"
    assert rel[:5].max() < 1e-2, rel[:5]
    assert rel[:10].max() < 3e-2, rel[:10]
    assert rel[:20].max() < 1e-1, rel


def test_ap3d_fp32_hip_equals_cpu_oracle_and_bf16_is_bounded(tmp_path, monkeypatch):
    from oracle import cpu_backend
    dev, cfg, over, fs, priors, batches = _setup(tmp_path, monkeypatch)
    model, traj = _train(cfg, priors, batches, "fp32", 1200)          # memorise the 8 images (20 ms per step)
    assert traj[-1] < 0.5 * traj[0], traj
    ops.set_precision("fp32")
    ap2, ap3, out = _evaluate(cfg, fs, model, "hip_fp32", tmp_path)
    assert ap3 > 20.0 and ap2 > 50.0, ("the memorised set must give meaningful detections", ap2, ap3)
    # ---- the float32 CPU oracle with the same weights on the same images
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    saved = {n: importlib.import_module(n).ops for n in cpu_backend.PATCHED}
    try:
        cpu_backend.install()
        cfg_cpu = syn.make_cfg(overrides=["MODEL.DEVICE", "cpu"] + over)
        ref = cpu_backend.attach(modeling.build_model(cfg_cpu, priors=priors))
        ref.load_state_dict(sd)
        ap2c, ap3c, outc = _evaluate(cfg_cpu, fs, ref, "cpu_oracle", tmp_path)
    finally:
        for n, o in saved.items():
            importlib.import_module(n).ops = o
    n_hip, n_cpu = sum(len(p["instances"]) for p in out), sum(len(p["instances"]) for p in outc)
    assert abs(ap3 - ap3c) <= 1e-3 and abs(ap2 - ap2c) <= 1e-3, ("AP3D / AP2D HIP-fp32 vs CPU oracle", ap3, ap3c, ap2, ap2c,
                                                                   n_hip, n_cpu)
    # ---- BASELINE configs[1] at size: ONE batch of 8 x 512 x 512 through the HIP path and through the CPU oracle with the same
    # weights -- detection by detection (same image, same class, 2D IoU > 0.9): scores, 2D boxes, 3D centres, dimensions and
    # the 8 corners.  float32 on both sides; what differs is the summation order of the f32 MFMA tiles vs ATen's CPU kernels
    # through ~60 layers, so the bound is 1e-3 relative (measured: printed), and the sets of detections must be the same
    # up to score-threshold / NMS ties (<= 2 % unmatched).
    ap2_8, ap3_8, out8 = _evaluate(cfg, fs, model, "hip_fp32_bs8", tmp_path, batch_size=8)
    assert abs(ap3_8 - ap3) <= 1e-3 and abs(ap2_8 - ap2) <= 1e-3, ("batch size must not change the detections", ap3_8, ap3)
    try:
        cpu_backend.install()
        _, _, outc8 = _evaluate(cfg_cpu, fs, ref, "cpu_oracle_bs8", tmp_path, batch_size=8)
    finally:
        for n, o in saved.items():
            importlib.import_module(n).ops = o
    worst, n_match, n_tot = _compare_detections(out8, outc8)
    print("bs=8 HIP-fp32 vs CPU oracle: matched %d of %d detections; worst relative deviations %s" % (n_match, n_tot, worst))
    assert n_match >= 0.98 * n_tot, (n_match, n_tot)
    assert all(v <= 1e-3 for v in worst.values()), worst
    # ---- the fp32x3 mode (float32 storage, contractions through the exact three-way bf16 split) on the same weights: held to
    # the SAME criterion as the default mode -- AP2D / AP3D within 1e-3 of the float32 CPU oracle
    prev = ops.set_precision("fp32x3")
    try:
        ap2x, ap3x, outx = _evaluate(cfg, fs, model, "hip_fp32x3", tmp_path)
    finally:
        ops.set_precision(prev)
    print(f"AP3D fp32x3 {ap3x:.4f} (cpu {ap3c:.4f}); AP2D fp32x3 {ap2x:.4f} (cpu {ap2c:.4f})")
    assert abs(ap3x - ap3c) <= 1e-3 and abs(ap2x - ap2c) <= 1e-3, ("AP3D / AP2D HIP-fp32x3 vs CPU oracle", ap3x, ap3c, ap2x, ap2c)
    worstx, n_mx, n_tx = _compare_detections(outx, outc)
    print("fp32x3 vs CPU oracle: matched %d of %d detections; worst relative deviations %s" % (n_mx, n_tx, worstx))
    assert n_mx >= 0.98 * n_tx and all(v <= 1e-3 for v in worstx.values()), (n_mx, n_tx, worstx)
    # ---- the bf16 fast mode on the same weights: measured deviation, stated bound
    prev = ops.set_precision("bf16")
    try:
        ap2b, ap3b, _ = _evaluate(cfg, fs, model, "hip_bf16", tmp_path)
    finally:
        ops.set_precision(prev)
    print(f"AP3D fp32 {ap3:.4f} cpu {ap3c:.4f} bf16 {ap3b:.4f}; AP2D fp32 {ap2:.4f} cpu {ap2c:.4f} bf16 {ap2b:.4f}")
    assert abs(ap3b - ap3) <= 5.0 and abs(ap2b - ap2) <= 5.0, (ap3b, ap3, ap2b, ap2)
