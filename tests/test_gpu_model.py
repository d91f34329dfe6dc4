"""GPU: the full Cube R-CNN DLA34-FPN train step and inference run end to end on the HIP kernels."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def built():
    bt = importlib.import_module("bench_train")
    return bt.build(DEV, seed=0)


def test_train_steps_run_and_learn(built):
    cfg, model, opt, syn, solver = built
    d2 = importlib.import_module("3dod_amd.d2lite")
    step = solver.TrainStep(cfg, model, opt, world_size=1)
    batch = syn.make_batch(2, 7)
    p0 = opt.flat_p.clone()
    totals = []
    with d2.EventStorage(0):
        for _ in range(6):
            step(batch)
            rep = step.report()
            totals.append(rep["total_loss"])
    assert all(t == t and abs(t) < 1e4 for t in totals), totals
    assert rep["iterations_explode"] == 0, rep
    assert not torch.equal(p0, opt.flat_p)
    expected = {"BoxHead/loss_cls", "BoxHead/loss_box_reg", "Cube/loss_dims", "Cube/loss_xy", "Cube/loss_z",
                "Cube/loss_pose", "Cube/loss_joint", "Cube/uncert", "rpn/cls", "rpn/loc"}
    assert expected <= set(rep.keys()), rep.keys()
    assert totals[-1] < totals[0], totals           # same batch 6 times: the loss goes down


def test_inference_runs(built):
    cfg, model, opt, syn, solver = built
    model.eval()
    thr = model.roi_heads.box_predictor.test_score_thresh
    model.roi_heads.box_predictor.test_score_thresh = -1.0     # barely-trained weights: keep detections alive
    try:
        with torch.no_grad():
            out = model(syn.make_batch(2, 9, with_gt=False))
    finally:
        model.roi_heads.box_predictor.test_score_thresh = thr
        model.train()
    assert len(out) == 2
    for o in out:
        inst = o["instances"]
        n = len(inst)
        assert 0 < n <= cfg.TEST.DETECTIONS_PER_IMAGE
        assert inst.pred_bbox3D.shape == (n, 8, 3) and inst.pred_pose.shape == (n, 3, 3)
        assert inst.pred_dimensions.shape == (n, 3) and inst.pred_center_cam.shape == (n, 3)
        assert inst.scores_full.shape == (n, cfg.MODEL.ROI_HEADS.NUM_CLASSES)
        assert torch.isfinite(inst.pred_bbox3D).all()


# ---------------------------------------------------------------------------------------------
# parity of the dense path: HIP (bf16 storage, f32 accumulate) vs float32 oracles
# ---------------------------------------------------------------------------------------------
def _rel(a, b):
    a = a.float().cpu(); b = b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12)), float((a - b).abs().max() / (b.abs().max() + 1e-12))


def test_dla_trunk_matches_reference_golden(golden_dir):
    """same seed -> same weights as the reference's DLA (tests/test_dla_weights.py); features vs the reference's
    own float32 forward (tests/golden/make_golden_dla.py).

    A randomly initialised 39-layer conv+BN+ReLU stack amplifies perturbations (measured: per-module error of the
    HIP kernels is ~1e-4 relative L2, see test_modules_in_isolation; end to end it grows ~1.15x per layer), so
      (a) every STAGE is checked in isolation, fed with the reference's own input for that stage (<= 9 layers deep),
      (b) the end-to-end chain only gets a loose bound."""
    import os
    import numpy as np
    dla = importlib.import_module("3dod_amd.cubercnn.modeling.backbone.dla")
    fpn = importlib.import_module("3dod_amd.cubercnn.modeling.backbone.fpn")
    g = np.load(os.path.join(golden_dir, "dla34_trunk.npz"), allow_pickle=False)
    torch.manual_seed(int(g["seed"]))
    net = fpn.to_channels_last(dla.dla34(pretrained=False)).to(DEV).train()
    to_dev = lambda a: torch.tensor(a).permute(0, 2, 3, 1).contiguous().to(DEV).to(torch.bfloat16)
    x = torch.tensor(g["x"])
    xh = torch.cat([x.permute(0, 2, 3, 1), torch.zeros(x.shape[0], x.shape[2], x.shape[3], 5)], 3)
    xh = xh.to(DEV).to(torch.bfloat16).contiguous()
    with torch.no_grad():
        b = net.base_layer(xh)
        l1 = net.level1(net.level0(b))
        l2 = net.level2(l1); l3 = net.level3(l2); l4 = net.level4(l3); l5 = net.level5(l4)
        iso = {"p2": net.level2(to_dev(g["level1"])), "p3": net.level3(to_dev(g["p2"])),
               "p4": net.level4(to_dev(g["p3"])), "p5": net.level5(to_dev(g["p4"]))}
    for name, got in iso.items():                                  # (a) stage by stage on the reference's inputs
        l2e, mx = _rel(got.permute(0, 3, 1, 2), torch.tensor(g[name]))
        assert l2e < 4e-2, ("isolated", name, l2e, mx)
    tol = {"base": 1e-2, "level1": 2e-2, "p2": 3e-2, "p3": 0.1, "p4": 0.25, "p5": 0.4}
    for name, got in (("base", b), ("level1", l1), ("p2", l2), ("p3", l3), ("p4", l4), ("p5", l5)):   # (b)
        l2e, mx = _rel(got.permute(0, 3, 1, 2), torch.tensor(g[name]))
        assert l2e < tol[name], (name, l2e, mx)


def test_modules_in_isolation(built, precision):
    """every conv module of the trunk, the FPN and the RPN head: the HIP result on the module's actual GPU inputs
    vs the float32 oracle on the same inputs.  This is the arithmetic check.
      fp32 (reference precision): relative L2 <= 2e-5, max-norm <= 2e-4 (summation order only);
      bf16 (fast mode) against the bf16-emulating oracle: relative L2 <= 2e-3, max-norm <= 2e-2 (one bf16 ulp at a few
      elements)."""
    from oracle import cpu_backend
    emu = precision == "bf16"
    t_l2, t_mx = (2e-3, 2e-2) if emu else (2e-5, 2e-4)
    cfg, model, opt, syn, solver = built
    dla = importlib.import_module("3dod_amd.cubercnn.modeling.backbone.dla")
    modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
    batch = syn.make_batch(2, 21, with_gt=False)
    rec = {}

    def mk(name):
        def hook(m, inp, out):
            rec[name] = ([i.detach().float().cpu() for i in inp if torch.is_tensor(i)], out.detach().float().cpu())
        return hook
    hooks = [m.register_forward_hook(mk(n)) for n, m in model.backbone.bottom_up.named_modules()
             if isinstance(m, (dla.BasicBlock, dla.Root, dla._Project, dla._ConvLevel))]
    bu = {}
    hooks.append(model.backbone.bottom_up.register_forward_hook(lambda m, i, o: bu.update(o)))
    model.train()
    with torch.no_grad():
        images, x = model.preprocess_image(batch)
        feats = model.backbone(x)       # (BN statistics use float atomics: a second run is not bitwise identical)
        pg = model.proposal_generator
        logits, deltas = pg.rpn_head([feats[f] for f in pg.in_features])
    for h in hooks:
        h.remove()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    saved = {n: importlib.import_module(n).ops for n in cpu_backend.PATCHED}
    try:
        cpu_backend.install()
        cpu_backend.EMULATE_BF16 = emu
        ref = modeling.build_model(syn.make_cfg(overrides=["MODEL.DEVICE", "cpu", "VIS_PERIOD", 0, "log", False]))
        ref.load_state_dict(sd)
        ref.train()
        mods = dict(ref.backbone.bottom_up.named_modules())
        worst = 0.0
        with torch.no_grad():
            for name, (inp, out) in rec.items():
                l2e, mx = _rel(mods[name](*inp), out)
                worst = max(worst, l2e)
                assert l2e < t_l2 and mx < t_mx, (name, l2e, mx)
            # FPN on the GPU's bottom-up features, RPN head on the GPU's FPN features
            class _BU(torch.nn.Module):
                def forward(self, x):
                    return {k: v.float().cpu() for k, v in bu.items()}
            real_bu = ref.backbone.bottom_up
            ref.backbone.bottom_up = _BU()
            f_ref = ref.backbone(None)
            ref.backbone.bottom_up = real_bu
            for k in feats:
                l2e, mx = _rel(feats[k], f_ref[k])
                assert l2e < 1.5 * t_l2 and mx < 1.5 * t_mx, (k, l2e, mx)
            lg, dl = ref.proposal_generator.rpn_head([feats[f].float().cpu() for f in pg.in_features])
            for a, b in zip(list(logits) + list(deltas), list(lg) + list(dl)):
                l2e, mx = _rel(a, b)
                assert l2e < 1.5 * t_l2 and mx < 1.5 * t_mx, (l2e, mx)
    finally:
        cpu_backend.EMULATE_BF16 = False
        for n, o in saved.items():
            importlib.import_module(n).ops = o


def test_graphed_dense_region_equals_eager(built):
    """HIP-graph replay of trunk+FPN+RPN head (forward AND backward) reproduces the eager result: same features,
    same flat gradient for a given upstream gradient."""
    cfg, model, opt, syn, solver = built
    batch = syn.make_batch(2, 33, with_gt=False)
    model.train()
    pg = model.proposal_generator

    def eager():
        opt.zero_grad()
        images, x = model.preprocess_image(batch)
        feats = model.backbone(x)
        logits, deltas = pg.rpn_head([feats[f] for f in pg.in_features])
        return feats, logits, deltas
    feats, logits, deltas = eager()
    loss = sum((f.float() ** 2).mean() for f in feats.values()) + sum(l.mean() for l in logits) + sum((d ** 2).mean() for d in deltas)
    loss.backward()
    opt.collect_grads()
    g_eager = opt.flat_g.clone()
    f_eager = {k: v.detach().clone() for k, v in feats.items()}
    runner = model.enable_graphs(batch)
    try:
        opt.zero_grad()
        images, u8 = model._stack_images(batch)
        feats2, ys2 = runner(u8)                          # the RPN head's raw per-level outputs (B,H,W,16): [A logits | 4A deltas | pad]
        A = pg.rpn_head.num_anchors
        ys2 = pg.rpn_head.level_views(ys2, [feats2[f] for f in pg.in_features])      # (the levels arrive stacked in one map)
        logits2 = [y[..., :A].reshape(y.shape[0], -1) for y in ys2]
        deltas2 = [y[..., A:5 * A].reshape(y.shape[0], -1, 4) for y in ys2]
        loss2 = sum((f.float() ** 2).mean() for f in feats2.values()) + sum(l.mean() for l in logits2) + sum((d ** 2).mean() for d in deltas2)
        loss2.backward()
        opt.collect_grads()
        for k in f_eager:
            assert torch.equal(f_eager[k], feats2[k]), k
        # atomics in the weight-gradient / BN kernels reorder float sums: equal to rounding, not bitwise
        rel = float((opt.flat_g - g_eager).norm() / g_eager.norm())
        assert rel < 1e-3, rel
        # replay again after a parameter update: the graphs must see the new weights
        opt.flat_p.mul_(1.01)
        importlib.import_module("3dod_amd.hipops").bump_weight_epoch()
        feats3, _ = runner(u8)
        assert not torch.equal(feats3["p2"], f_eager["p2"])
        f4, _, _ = eager()
        assert torch.equal(f4["p2"], feats3["p2"])
    finally:
        model._graphed = None
        opt.flat_p.div_(1.01)
        opt.zero_grad()


def test_fused_cube_kernel_matches_reference_golden(golden_dir):
    """cr_cube_loss_fwd/_bwd (K15/K16) inside ROIHeads3D._forward_cube on the GPU vs the golden vectors produced
    by the REFERENCE's own _forward_cube: losses 1e-5 rel, gradients w.r.t. the head outputs 5e-4, corners 1e-4."""
    import os
    import numpy as np
    tc = importlib.import_module("tests.test_cubehead_golden") if False else None
    d2 = importlib.import_module("3dod_amd.d2lite")
    syn = importlib.import_module("3dod_amd.synthetic")
    modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
    util = importlib.import_module("3dod_amd.cubercnn.util.math_util")
    g = np.load(os.path.join(golden_dir, "cubehead_train.npz"), allow_pickle=False)
    cfg = syn.make_cfg(overrides=["MODEL.DEVICE", "cuda:0", "VIS_PERIOD", 0, "log", False])
    shapes = {f"p{l}": d2.ShapeSpec(channels=256, stride=2 ** l) for l in range(2, 7)}
    from oracle import list_path
    heads = list_path.install_heads(modeling.build_roi_heads(cfg, shapes).to(DEV).train())     # list-shaped caller of the kernel pair
    n_per = g["n_per"].tolist()
    T = lambda k: torch.tensor(g[k]).to(DEV)
    insts = []
    for i in range(len(n_per)):
        inst = d2.Instances((512, 512))
        sp = lambda k: T(k).split(n_per)[i]
        inst.proposal_boxes = d2.Boxes(sp("proposal_boxes")); inst.pred_boxes = d2.Boxes(sp("pred_boxes"))
        inst.gt_classes = sp("gt_classes"); inst.gt_boxes3D = sp("gt_boxes3D"); inst.gt_poses = sp("gt_poses")
        insts.append(inst)
    leaves = {k: T("in_" + k).requires_grad_(True) for k in ("deltas", "z", "dims", "pose6", "uncert")}
    n = leaves["z"].shape[0]
    pose = util.rotation_6d_to_matrix(leaves["pose6"].view(-1, 6)).view(n, -1, 3, 3)
    heads.priors_dims_per_cat.data = T("priors")

    class _Fake(torch.nn.Module):
        def __init__(self, fn):
            super().__init__(); self.fn = fn

        def forward(self, *a):
            return self.fn(*a)
    heads.cube_pooler = _Fake(lambda feats, boxes: torch.zeros(n, 4, device=DEV))
    heads.cube_head = _Fake(lambda x: (leaves["deltas"], leaves["z"], leaves["dims"], pose, leaves["uncert"]))
    Ks = [torch.tensor(k) for k in g["Ks"]]
    with d2.EventStorage(0):
        pred, losses = heads._forward_cube({f: None for f in heads.in_features}, insts, Ks, [(512, 512)] * 3,
                                           [float(r) for r in g["ratios"]])
    for k, v in losses.items():
        ref = float(g["loss_" + k.replace("/", "_")])
        assert abs(float(v) - ref) <= 2e-5 * max(1.0, abs(ref)), (k, float(v), ref)
    sum(losses.values()).backward()
    for k, leaf in leaves.items():
        np.testing.assert_allclose(leaf.grad.cpu().numpy(), g["grad_" + k], rtol=5e-4, atol=5e-6, err_msg=k)
    for f in ("pred_bbox3D", "pred_center_cam", "pred_center_2D", "pred_dimensions", "pred_pose", "scores"):
        got = torch.cat([i.get(f) for i in pred]).detach().cpu().numpy()
        np.testing.assert_allclose(got, g["out_" + f], rtol=1e-4, atol=2e-5, err_msg=f)


def test_train_steps_with_graphs_match_eager():
    """the HIP-graph path over several optimizer steps and DIFFERENT batches: no skipped step, and the loss trace
    follows the eager trace (same seeds; wgrad float atomics make it equal to rounding, not bitwise)."""
    bt = importlib.import_module("bench_train")
    d2 = importlib.import_module("3dod_amd.d2lite")

    def run(graphs):
        cfg, model, opt, syn, solver = bt.build(DEV, seed=0)
        batches = [syn.make_batch(2, 500 + i) for i in range(3)]
        if graphs:
            model.enable_graphs(batches[0])
            opt.zero_grad()
        step = solver.TrainStep(cfg, model, opt)
        torch.manual_seed(123)                      # same sampling streams in both runs
        tr = []
        with d2.EventStorage(0):
            for i in range(5):
                step(batches[i % 3])
                r = step.report()
                tr.append(r["total_loss"])
        return tr, r["iterations_explode"], float(opt.flat_g.abs().max())
    te, xe, ge = run(False)
    tg, xg, gg = run(True)
    assert xe == 0 and xg == 0, (xe, xg)
    assert all(t == t and t < 100 for t in tg), tg
    assert gg < 1e4 and ge < 1e4
    assert abs(tg[0] - te[0]) < 1e-3 * abs(te[0]), (tg, te)          # first step: identical weights and samples
    assert abs(tg[1] - te[1]) < 0.05 * abs(te[1]), (tg, te)


def test_whole_step_graph_trains_like_eager():
    """GraphedTrainStep (two HIP graphs per step) vs the eager TrainStep on the same data: first-step losses equal,
    no skipped steps, parameters move, state restored after capture warm-up."""
    bt = importlib.import_module("bench_train")
    d2 = importlib.import_module("3dod_amd.d2lite")

    def run(graph):
        cfg, model, opt, syn, solver = bt.build(DEV, seed=0, lr=0.0025)
        batches = [syn.make_batch(2, 700 + i) for i in range(3)]
        for b in batches:
            for d in b:
                d["image"] = d["image"].to(DEV); d["instances"] = d["instances"].to(DEV)
        p0 = opt.flat_p.clone()
        with d2.EventStorage(0):
            step = solver.GraphedTrainStep(cfg, model, opt, batches[0]) if graph else solver.TrainStep(cfg, model, opt)
            assert torch.equal(p0, opt.flat_p), "capture warm-up must not change the parameters"
            tr = []
            for i in range(6):
                step(batches[i % 3])
                tr.append(step.report())
        return tr, opt.flat_p.clone(), p0
    te, pe, _ = run(False)
    tg, pg, p0 = run(True)
    assert tg[-1]["iterations_explode"] == 0 and te[-1]["iterations_explode"] == 0
    assert not torch.equal(pg, p0)
    # same weights at step 0; the sampling RNG streams differ (graph-safe philox offsets), so compare the deterministic
    # part tightly and the sampled losses loosely
    for k in ("rpn/cls", "rpn/loc", "BoxHead/loss_cls"):
        assert abs(tg[0][k] - te[0][k]) < 0.15 * abs(te[0][k]) + 1e-3, (k, tg[0][k], te[0][k])
    assert abs(tg[0]["total_loss"] - te[0]["total_loss"]) < 0.05 * te[0]["total_loss"]
    assert all(t["total_loss"] == t["total_loss"] and t["total_loss"] < 50 for t in tg)
    assert tg[-1]["total_loss"] < tg[0]["total_loss"]


def test_two_segment_backward_equals_single_graph(built):
    """GraphedDense(split_backward=True): the backward captured as [RPN head, FPN, level5] | [level4 ... stem] gives the
    flat gradient of the single backward graph (same upstream gradient; float sums reorder through atomics only), every
    parameter of the region belongs to exactly one segment, and the first segment holds most of the region's parameters."""
    cfg, model, opt, syn, solver = built
    batch = syn.make_batch(2, 34, with_gt=False)
    model.train()
    pg = model.proposal_generator
    images, u8 = model._stack_images(batch)
    A = pg.rpn_head.num_anchors

    def run(split):
        runner = model.enable_graphs(batch, split_backward=split)
        assert (runner.bwd_graph2 is not None) == split
        opt.zero_grad()
        feats, ys = runner(u8)
        logits = [y[..., :A].reshape(y.shape[0], -1) for y in ys]
        deltas = [y[..., A:5 * A].reshape(y.shape[0], -1, 4) for y in ys]
        loss = sum((f.float() ** 2).mean() for f in feats.values()) + sum(l.mean() for l in logits) + sum((d ** 2).mean() for d in deltas)
        loss.backward()
        opt.collect_grads()
        return runner, opt.flat_g.clone()
    try:
        _, g1 = run(False)
        runner, g2 = run(True)
        rel = float((g2 - g1).norm() / g1.norm())
        assert rel < 1e-4, rel
        first, second = runner.segment_params(True), runner.segment_params(False)
        region = [p for m in (model.backbone, pg.rpn_head) for p in m.parameters() if p.requires_grad]
        assert len(first) + len(second) == len(region) and not ({id(p) for p in first} & {id(p) for p in second})
        n1, n2 = sum(p.numel() for p in first), sum(p.numel() for p in second)
        assert n1 > n2 > 1_000_000, (n1, n2)
        # every gradient of the second segment is non-zero only through the second graph
        r1 = solver.param_ranges(first, opt); r2 = solver.param_ranges(second, opt)
        assert all(float(g2[a:b].abs().max()) > 0 for a, b in r1) and all(float(g2[a:b].abs().max()) > 0 for a, b in r2)
    finally:
        model._graphed = None


def test_whole_step_graph_run_ahead_lr_schedule_and_gt_overflow():
    """GraphedTrainStep with the host running ahead of the device (the default: no per-step sync) at the bench's batch
    size -- the mode that faulted in round 1 (garbage sampling indices from aliased `.contiguous()` temporaries whose
    pointers were read inline; every wrapper now owns its temporaries until the launch, hipops._Args).  Also: the
    learning-rate schedule reaches the captured update through a device scalar (lr_scale = 0 freezes the parameters),
    cached compute copies of the weights follow the graph's updates, and a batch with more ground-truth rows than the
    captured buffers hold re-captures instead of failing."""
    bt = importlib.import_module("bench_train")
    d2 = importlib.import_module("3dod_amd.d2lite")
    ops = importlib.import_module("3dod_amd.hipops")
    cfg, model, opt, syn, solver = bt.build(DEV, seed=0, lr=0.0025)
    batches = [syn.make_batch(4, 1234 + i) for i in range(4)]
    for b in batches:
        for d in b:
            d["image"] = d["image"].to(DEV); d["instances"] = d["instances"].to(DEV)
    with d2.EventStorage(0):
        step = solver.GraphedTrainStep(cfg, model, opt, batches[0])
        assert not step.sync_each_step and step.G >= 32
        for i in range(40):                                   # enqueued back to back: nothing here waits for the device
            step(batches[i % 4])
        rep = step.report()                                   # the one sync
        assert rep["iterations_explode"] == 0 and rep["iterations_success"] == 40
        assert rep["total_loss"] == rep["total_loss"] and rep["total_loss"] < 50
        # --- schedule: factor 0 -> the captured update leaves the parameters alone; factor 1 moves them again
        p0 = opt.flat_p.clone()
        opt.lr_scale = 0.0
        step(batches[0]); step(batches[1])
        # (weight decay and momentum are scaled by the learning rate too: p -= lr * m)
        assert torch.equal(opt.flat_p, p0), "lr_scale = 0 must freeze the parameters of the captured update"
        opt.lr_scale = 1.0
        step(batches[2])
        assert not torch.equal(opt.flat_p, p0)
        # --- compute copies of the weights used by eager code after graph steps are current (weight epoch bumped)
        w = next(p for p in model.backbone.parameters() if p.dim() == 4 and p.shape[1] >= 16)
        wb, _ = ops.prepared_weights(w, False, torch.bfloat16)
        assert torch.equal(wb.float().view(-1), w.detach().permute(0, 2, 3, 1).reshape(-1).to(torch.bfloat16).float())
        # --- more ground-truth rows than the captured buffers hold
        big = syn.make_batch(4, 99, min_obj=40, max_obj=40)
        for d in big:
            d["image"] = d["image"].to(DEV); d["instances"] = d["instances"].to(DEV)
        step(big)
        assert step.G >= 40
        rep = step.report()
        assert rep["total_loss"] == rep["total_loss"] and rep["iterations_explode"] == 0


def test_train_step_comm_protocol_single_rank_group(built):
    """the multi-rank protocol (side stream, RoI-head FC gradients all-reduced while the captured trunk backward runs,
    the rest after backward) exercised on a 1-rank RCCL group: same code path as N > 1, all-reduce = identity."""
    import torch.distributed as dist
    cfg, model, opt, syn, solver = built
    d2 = importlib.import_module("3dod_amd.d2lite")
    model.train()
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29731", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        step = solver.TrainStep(cfg, model, opt, world_size=1, force_comm=True)
        n = opt.flat_g.numel()
        assert step.early_ranges and sum(b - a for a, b in step.early_ranges) > 20_000_000      # the four big FC weights
        cover = sorted(step.early_ranges + step.late_ranges)
        assert cover[0][0] == 0 and cover[-1][1] == n and all(x[1] == y[0] for x, y in zip(cover, cover[1:]))
        batch = syn.make_batch(4, 9)
        model.enable_graphs(batch, split_backward=True)      # two backward graphs, as bench.py --gpus N > 1 runs
        opt.zero_grad()
        p0 = opt.flat_p.clone()
        with d2.EventStorage(0):
            for _ in range(3):
                step(batch)
                assert step._early_done                      # the hook fired before the backward graph
                assert step._mid_done                        # ... and the second one between the two backward segments
        # the three phases partition the flat gradient
        cover = sorted(step.early_ranges + step.mid_ranges + step.late_after_mid)
        assert cover[0][0] == 0 and cover[-1][1] == n and all(x[1] == y[0] for x, y in zip(cover, cover[1:]))
        assert sum(b - a for a, b in step.mid_ranges) > 10_000_000       # RPN head + FPN + DLA level5
        assert step._rest_ranges() == step.late_after_mid                # the third phase = exactly what the first two left
        rep = step.report()
        assert rep["total_loss"] == rep["total_loss"] and rep["iterations_explode"] == 0, rep
        assert not torch.equal(p0, opt.flat_p)
    finally:
        model._graphed = None
        dist.destroy_process_group()


def test_batched_fast_rcnn_inference_equals_per_image():
    """the batched post-processing (one grouped NMS over image x class groups) returns exactly what the reference-shaped
    per-image function returns (fast_rcnn.py:57-116)."""
    fr = importlib.import_module("3dod_amd.cubercnn.modeling.roi_heads.fast_rcnn")
    g = torch.Generator().manual_seed(4)
    K, shapes, sizes = 6, [(200, 240), (256, 256), (180, 300)], [300, 257, 120]
    boxes, scores = [], []
    for n, (h, w) in zip(sizes, shapes):
        ctr = torch.rand(n, K, 2, generator=g) * torch.tensor([w, h])
        wh = torch.rand(n, K, 2, generator=g) * 80 + 4
        b = torch.cat([ctr - wh / 2, ctr + wh / 2], -1).reshape(n, K * 4)
        s = torch.softmax(torch.randn(n, K + 1, generator=g) * 2, 1)
        b[3, 0] = float("nan")                                   # a non-finite row is dropped
        boxes.append(b.to(DEV)); scores.append(s.to(DEV))
    got, got_rows = fr.fast_rcnn_inference(boxes, scores, shapes, 0.05, 0.5, 40)
    for i in range(3):
        ref, ref_rows = fr.fast_rcnn_inference_single_image(boxes[i], scores[i], shapes[i], 0.05, 0.5, 40)
        # the per-image function compacts non-finite rows first: map its row indices back
        valid = torch.isfinite(boxes[i]).all(1) & torch.isfinite(scores[i]).all(1)
        ref_rows = valid.nonzero()[:, 0][ref_rows]
        assert len(ref) == len(got[i]) and 0 < len(ref) <= 40
        assert torch.equal(got[i].pred_boxes.tensor, ref.pred_boxes.tensor) and torch.equal(got[i].scores, ref.scores)
        assert torch.equal(got[i].pred_classes, ref.pred_classes) and torch.equal(got[i].scores_full, ref.scores_full)
        assert torch.equal(got_rows[i], ref_rows)


def test_fused_cube_inference_equals_torch_path(built):
    """the fused inference decode (cr_cube_decode_infer) against the reference-shaped torch expressions of oracle/cube_list.py
    on the same detections (the torch path is what tests/test_cubehead_golden.py pins to the reference)."""
    cfg, model, opt, syn, solver = built
    ops = importlib.import_module("3dod_amd.hipops")
    d2 = importlib.import_module("3dod_amd.d2lite")
    rh = model.roi_heads
    model.eval()
    try:
        batch = syn.make_batch(3, 17, with_gt=False)
        with torch.no_grad():
            images, x = model.preprocess_image(batch)
            feats = model.backbone(x)
            g = torch.Generator().manual_seed(2)
            dets = []
            for i, n in enumerate((7, 1, 12)):
                ctr = torch.rand(n, 2, generator=g) * 300 + 100
                wh = torch.rand(n, 2, generator=g) * 150 + 20
                inst = d2.Instances(images.image_sizes[i])
                inst.pred_boxes = d2.Boxes(torch.cat([ctr - wh / 2, ctr + wh / 2], 1).to(DEV))
                inst.pred_classes = torch.randint(0, rh.num_classes, (n,), generator=g).to(DEV)
                inst.scores = torch.rand(n, generator=g).to(DEV)
                dets.append(inst)
            Ks = [torch.FloatTensor(b["K"]) for b in batch]
            ratios = [1.0, 1.0, 1.0]
            dims = [tuple(s) for s in images.image_sizes]
            clone = lambda L: [d2.Instances(i.image_size, **{k: (v.clone() if torch.is_tensor(v) else d2.Boxes(v.tensor.clone()))
                                                             for k, v in i.get_fields().items()}) for i in L]
            fused = rh._forward_cube(feats, clone(dets), Ks, dims, ratios)
            from oracle import cube_list                       # the torch expressions (pinned to the reference on the CPU)
            ref = cube_list.forward_cube_list(rh, feats, clone(dets), Ks, dims, ratios)
        for a, b in zip(fused, ref):
            for k in ("scores", "pred_bbox3D", "pred_center_cam", "pred_center_2D", "pred_dimensions", "pred_pose"):
                x1, x2 = a.get(k).float(), b.get(k).float()
                assert x1.shape == x2.shape, k
                assert float((x1 - x2).abs().max()) <= 2e-2 * float(x2.abs().max()) + 1e-3, (k, float((x1 - x2).abs().max()))
    finally:
        model.train()


def test_eval_graph_matches_eager_inference(built):
    """inference through the forward-only HIP graph (trunk + FPN + RPN head) == eager inference on the same weights."""
    cfg, model, opt, syn, solver = built
    model.eval()
    thr = model.roi_heads.box_predictor.test_score_thresh
    model.roi_heads.box_predictor.test_score_thresh = -1.0
    try:
        batch = syn.make_batch(2, 23, with_gt=False)
        with torch.no_grad():
            eager = model(batch)
            model.enable_graphs_eval(batch)
            graphed = model(batch)
            graphed2 = model(syn.make_batch(2, 24, with_gt=False))       # new pixels through the same graph
        for a, b in zip(eager, graphed):
            ia, ib = a["instances"], b["instances"]
            assert len(ia) == len(ib) > 0
            assert torch.allclose(ia.pred_boxes.tensor, ib.pred_boxes.tensor, atol=1e-3)
            assert torch.allclose(ia.scores, ib.scores, atol=1e-4) and torch.equal(ia.pred_classes, ib.pred_classes)
            assert torch.allclose(ia.pred_bbox3D, ib.pred_bbox3D, atol=1e-3)
        assert len(graphed2) == 2 and not torch.equal(graphed2[0]["instances"].scores, graphed[0]["instances"].scores)
        # the BatchNorm folds live outside the graph (ops.refresh_folds): a running statistic / a scale changed after the
        # capture must reach the next replay
        ge = model._graphed_eval
        assert len(ge.folds) >= 30 and all(e["tag"] is not None for e in ge.folds)
        bn = model.backbone.bottom_up.level2.tree1.bn1
        old_var, old_w = bn.running_var.clone(), bn.weight.data.clone()
        with torch.no_grad():
            bn.running_var.mul_(1.7)
            bn.weight.data.mul_(0.8)
            changed = model(batch)
            cache, model._graphed_eval_cache = model._graphed_eval_cache, None
            eager_changed = model(batch)
            model._graphed_eval_cache = cache
            bn.running_var.copy_(old_var)
            bn.weight.data.copy_(old_w)
            back = model(batch)
        assert not torch.equal(changed[0]["instances"].scores, graphed[0]["instances"].scores)
        assert torch.allclose(changed[0]["instances"].scores, eager_changed[0]["instances"].scores, atol=1e-4)
        assert torch.equal(back[0]["instances"].scores, graphed[0]["instances"].scores)
        # per-shape cache: a second resolution is captured on first sight, the least recently used shape is dropped
        with torch.no_grad():
            model.enable_graphs_eval(max_shapes=2)
            small = syn.make_batch(1, 25, size=256, with_gt=False)
            out_small = model(small)
            assert {k[0] for k in model._graphed_eval_cache} == {(2, 3, 512, 512), (1, 3, 256, 256)}      # keys: (shape, precision)
            model._graphed_eval_cache.clear()
            eager_small = model(small)                                     # cache empty again -> re-captured; compare with eager
            model._graphed_eval_max = 0
            model._graphed_eval_cache.clear()
            eager_small = model(small)
            assert len(out_small[0]["instances"]) == len(eager_small[0]["instances"])
            assert torch.allclose(out_small[0]["instances"].scores, eager_small[0]["instances"].scores, atol=1e-4)
            model.enable_graphs_eval(max_shapes=2)
            model(small); model(syn.make_batch(1, 26, size=384, with_gt=False)); model(syn.make_batch(1, 27, size=320, with_gt=False))
            assert [k[0] for k in model._graphed_eval_cache] == [(1, 3, 384, 384), (1, 3, 320, 320)]
    finally:
        model._graphed_eval = None
        model._graphed_eval_cache, model._graphed_eval_max = None, 0
        model.roi_heads.box_predictor.test_score_thresh = thr
        model.train()


def test_eval_proposals_device_path_equals_list_path(built):
    """RPN.predict_proposals: fused device path (batched top-k, decode of the candidates only, grouped NMS) against the
    reference-shaped per-level path (decode everything, find_top_rpn_proposals) on the same head outputs."""
    cfg, model, opt, syn, solver = built
    ops = importlib.import_module("3dod_amd.hipops")
    model.eval()
    try:
        batch = syn.make_batch(2, 31, with_gt=False)
        with torch.no_grad():
            images, x = model.preprocess_image(batch)
            feats = model.backbone(x)
            pg = model.proposal_generator
            fl = [feats[f] for f in pg.in_features]
            anchors = pg.anchor_generator([(f.shape[1], f.shape[2]) for f in fl], DEV)
            logits, deltas = pg.rpn_head(fl)
            a = pg.predict_proposals(anchors, logits, deltas, images.image_sizes)
            f = ops.rpn_decode_select
            try:
                del ops.rpn_decode_select
                b = pg.predict_proposals(anchors, logits, deltas, images.image_sizes)
            finally:
                ops.rpn_decode_select = f
        for pa, pb in zip(a, b):
            assert len(pa) == len(pb) > 0
            assert torch.allclose(pa.objectness_logits, pb.objectness_logits, atol=1e-6)
            assert torch.allclose(pa.proposal_boxes.tensor, pb.proposal_boxes.tensor, atol=1e-3)
    finally:
        model.train()


def test_train_step_other_resolutions(built):
    """the kernels' tile / split / wave-group heuristics and the labelling path on shapes other than the benchmark's:
    384x384 images, and a batch of 300x448 crops (non-square, zero-padded to 320 rows for the FPN's size divisibility)."""
    cfg, model, opt, syn, solver = built
    d2 = importlib.import_module("3dod_amd.d2lite")
    model.train()
    model._graphed = None
    step = solver.TrainStep(cfg, model, opt, world_size=1)
    b1 = syn.make_batch(2, 41, size=384)
    b2 = syn.make_batch(2, 42, size=448)
    for d in b2:                                                    # crop to 300 x 448, keep the objects that survive
        d["image"] = d["image"][:, :300, :].contiguous()
        d["height"] = 300
        inst = d["instances"]
        keep = inst.gt_boxes.tensor[:, 1] < 280
        new = d2.Instances((300, 448))
        bx = inst.gt_boxes.tensor[keep].clone()
        bx[:, 3] = bx[:, 3].clamp(max=299)
        new.gt_boxes = d2.Boxes(bx)
        new.gt_classes = inst.gt_classes[keep]
        new.gt_boxes3D = inst.gt_boxes3D[keep]
        new.gt_poses = inst.gt_poses[keep]
        d["instances"] = new
    with d2.EventStorage(0):
        for b in (b1, b2, b1):
            step(b)
            rep = step.report()
            assert rep["total_loss"] == rep["total_loss"] and abs(rep["total_loss"]) < 1e4, rep
    assert rep["iterations_explode"] == 0


def test_image_without_objects_and_state_dict_roundtrip(built):
    """(i) a training batch in which one image has no ground-truth object (the reference's `if len(gt) == 0` branches);
    (ii) state_dict -> load_state_dict into a fresh model reproduces the eval outputs (the bf16 weight copies are keyed
    by a weight epoch that the load hook moves)."""
    cfg, model, opt, syn, solver = built
    d2 = importlib.import_module("3dod_amd.d2lite")
    modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
    model.train()
    model._graphed = None
    step = solver.TrainStep(cfg, model, opt, world_size=1)
    batch = syn.make_batch(2, 51)
    empty = d2.Instances((512, 512))
    empty.gt_boxes = d2.Boxes(torch.zeros(0, 4))
    empty.gt_classes = torch.zeros(0, dtype=torch.int64)
    empty.gt_boxes3D = torch.zeros(0, 9)
    empty.gt_poses = torch.zeros(0, 3, 3)
    batch[1]["instances"] = empty
    with d2.EventStorage(0):
        step(batch)
        rep = step.report()
    assert rep["total_loss"] == rep["total_loss"] and abs(rep["total_loss"]) < 1e4 and rep["iterations_explode"] == 0, rep
    # round trip
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    torch.manual_seed(123)
    fresh = modeling.build_model(cfg).eval()
    fresh.load_state_dict(sd)
    model.eval()
    thr = model.roi_heads.box_predictor.test_score_thresh
    model.roi_heads.box_predictor.test_score_thresh = fresh.roi_heads.box_predictor.test_score_thresh = -1.0
    try:
        tb = syn.make_batch(2, 52, with_gt=False)
        with torch.no_grad():
            a, b = model(tb), fresh(tb)
        for x, y in zip(a, b):
            ix, iy = x["instances"], y["instances"]
            assert len(ix) == len(iy) > 0
            assert torch.allclose(ix.scores, iy.scores, atol=1e-5) and torch.allclose(ix.pred_bbox3D, iy.pred_bbox3D, atol=1e-4)
    finally:
        model.roi_heads.box_predictor.test_score_thresh = thr
        model.train()
