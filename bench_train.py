"""Cube R-CNN DLA34-FPN train step benchmark (BASELINE.json metric): 4 synthetic 512x512 images per GPU,
forward + losses + backward + gradient all-reduce + SGD-momentum update, every step.

The headline line is measured in the REFERENCE'S precision: float32 activations / operands on the f32 MFMA
(the reference trains without autocast, tools/train_net.py:184-330), roofline against the 157.3 TFLOP/s f32 peak.
At N = 1 the same step is then re-measured in the bf16 fast mode and reported inside the same JSON line under
"bf16_mode" (labelled; it is NOT the headline: tests/test_gpu_precision_parity.py bounds its deviation)."""
import importlib
import json
import os
import time

import torch

# MI355X_MICROARCH.md: dense bf16 ~2.5 PFLOP/s; f32 MFMA 157.3 TFLOP/s.  fp32x3 (f32 operands split exactly into three
# bf16 values, six bf16 MFMAs per f32 multiply-add): the matrix-core bound is the bf16 peak / 6
MFMA_PEAK = {"bf16": 2500.0, "fp32": 157.3, "fp32x3": 2500.0 / 6.0}
PRECISION_TEXT = {
    "fp32": "float32 activations, weights and gradients, f32 MFMA (the reference's arithmetic)",
    "fp32x3": "float32 activations, weights and gradients; contractions on the bf16 matrix cores with every f32 operand split "
              "exactly into three bf16 values, six products per term, f32 accumulate (error vs float64 equal to the f32 "
              "MFMA path: tests/test_gpu_convops_x3.py); layers outside the split kernels on the f32 MFMA",
    "bf16": "bf16 activations / operands, f32 accumulate and parameters",
}
MFMA_PEAK_TFLOPS = MFMA_PEAK["bf16"]            # (name kept for scripts/)
IMS_PER_GPU = 4
TRAIN_GFLOP_PER_IMAGE = 309.0  # BASELINE.md section 2 (fwd 51.5 GMAC x 2 x 3)


def build(dev, seed=0, lr=None, world=1, config=None, extra=()):
    # linear LR scaling rule of the reference (README.md:230-245): configs/Base.yaml's 0.02 is for 32 images/batch
    if lr is None:
        lr = 0.02 * IMS_PER_GPU * world / 32.0
    syn = importlib.import_module("3dod_amd.synthetic")
    modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
    solver = importlib.import_module("3dod_amd.cubercnn.solver")
    # CR_CONFIG selects another model config of configs/ (e.g. cubercnn_ResNet34_FPN.yaml); the default is the
    # BASELINE one (Base_Omni3D.yaml = DLA34-FPN)
    cfg_file = config or os.environ.get("CR_CONFIG")
    if cfg_file and not os.path.isabs(cfg_file):
        cfg_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs", cfg_file)
    cfg = syn.make_cfg(cfg_file, overrides=["MODEL.DEVICE", str(dev), "VIS_PERIOD", 0, "log", False,
                                  "SOLVER.IMS_PER_BATCH", 32, "SOLVER.BASE_LR", lr] + list(extra))
    torch.manual_seed(seed)
    model = modeling.build_model(cfg)
    model.train()
    opt = solver.build_optimizer(cfg, model)
    return cfg, model, opt, syn, solver


def bench_train(args, rank, world, dev):
    ops = importlib.import_module("3dod_amd.hipops")
    main_prec = ops.precision()                          # fp32 unless CR_PRECISION=bf16 was asked for explicitly
    res = _bench_train_mode(args, rank, world, dev, main_prec)
    notes = {"fp32x3": "same step, float32 storage, contractions through the exact three-way bf16 operand split (float32 "
                       "accuracy, opt-in: CR_PRECISION=fp32x3); not the headline",
             "bf16": "same step with bf16 activations / operands (fast mode, opt-in); not the headline; deviation from "
                     "the fp32 mode bounded by tests/test_gpu_precision_parity.py"}
    if world == 1 and os.environ.get("CR_BENCH_DO_TRAIN", "1") == "1" and not getattr(args, "lean", False):
        try:
            res["do_train_loop"] = bench_do_train(args, rank, world, dev)
        except Exception as e:           # never lose the headline to a secondary measurement
            res["do_train_loop"] = {"error": f"{type(e).__name__}: {e}"}
    if main_prec == "fp32" and world == 1 and os.environ.get("CR_BENCH_BF16", "1") == "1" and not getattr(args, "lean", False):
        for extra in ("fp32x3", "bf16"):
            prev = ops.set_precision(extra)
            try:
                fast = _bench_train_mode(args, rank, world, dev, extra, is_main=False)
                res[extra + "_mode"] = {k: fast[k] for k in ("value", "unit", "ms_per_step", "dtype", "roofline")}
                res[extra + "_mode"]["note"] = notes[extra]
                res[extra + "_mode"]["final_loss"] = fast["config"]["final_loss"]
                res[extra + "_mode"]["valid"] = fast["config"]["valid"]
            finally:
                ops.set_precision(prev)
    return res


def _bench_train_mode(args, rank, world, dev, prec, is_main=True):
    import bench as B
    peak = MFMA_PEAK[prec]
    cfg, model, opt, syn, solver = build(dev, world=world)
    if world > 1:
        import torch.distributed as dist
        dist.broadcast(opt.flat_p, 0)                 # DDP wrap-time parameter broadcast
        importlib.import_module("3dod_amd.hipops").bump_weight_epoch()      # parameters changed outside the optimizer
    d2 = importlib.import_module("3dod_amd.d2lite")
    batches = [syn.make_batch(IMS_PER_GPU, 1234 + rank * 1000 + i) for i in range(4)]
    for b in batches:                                   # inputs resident in HBM before the timed region
        for d in b:
            d["image"] = d["image"].to(dev)
            d["instances"] = d["instances"].to(dev)
    # launch mode = the product's: solver.make_train_step is what do_train (tools/train_net.py) runs -- TrainStep with the
    # dense region replayed from per-shape HIP graphs (CR_GRAPHS=dense, the default) | step (whole-step graphs, opt-in) | none
    mode = os.environ.get("CR_GRAPHS", "dense")
    step = None
    if mode == "step":
        try:
            with d2.EventStorage(0):
                step = solver.GraphedTrainStep(cfg, model, opt, batches[0], world_size=world)
        except Exception as e:                          # never lose the number to a capture problem
            import sys
            print(f"[bench] whole-step graph capture failed ({type(e).__name__}: {e}); falling back", file=sys.stderr, flush=True)
            mode = "dense"
    if step is None:
        if mode == "step":
            mode = "dense"
        # data-parallel runs: two backward graphs, the first segment's gradients are all-reduced under the second
        step = solver.make_train_step(cfg, model, opt, world_size=world)
        if os.environ.get("CR_BWD_SPLIT") == "force" and mode == "dense":            # A/B: the two-segment backward on one GPU
            model.enable_graphs(None, split_backward=True, max_shapes=8)
        with d2.EventStorage(0):
            step(batches[0])                            # first sight of the batch shape: the dense region is captured here
    trace = os.environ.get("CR_TRACE") == "1"
    def note(msg):
        if trace:
            import sys
            print(f"[bench-trace] {msg}", file=sys.stderr, flush=True)
    note(f"built mode={mode} precision={prec}")
    with d2.EventStorage(0):
        for i in range(args.warmup):
            step(batches[i % len(batches)])
            note(f"warmup {i} enqueued")
        B.barrier(world)
        note("warmup done")
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(batches[i % len(batches)])
            note(f"step {i} enqueued")
        B.barrier(world)
        note("timed loop done")
        dt = B.max_over_ranks(time.perf_counter() - t0, world, dev)
        rep = step.report()
    ims = IMS_PER_GPU * world * args.steps / dt
    achieved_tf = TRAIN_GFLOP_PER_IMAGE * IMS_PER_GPU / (dt / args.steps) / 1e3
    comm = {"world_size": world, "backend": "none (single process)"}
    if world > 1:
        # the step's one real exchange on its own: all-reduce of the flat float32 gradient over RCCL, so that a scaling
        # run explains itself (bus bandwidth = 2 (n-1)/n x bytes / time, the ring's per-link figure)
        import torch.distributed as dist
        buf = torch.empty_like(opt.flat_g)
        for _ in range(2):
            dist.all_reduce(buf)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(5):
            dist.all_reduce(buf)
        torch.cuda.synchronize(dev)
        ar = B.max_over_ranks((time.perf_counter() - t1) / 5, world, dev)
        nbytes = buf.numel() * 4
        # data-parallel invariant: every rank holds the same parameters after the timed steps (same updates from the same
        # all-reduced gradients); max - min over ranks of a strided parameter checksum must be exactly 0
        chk = opt.flat_p[::997].double().sum().reshape(1)
        hi, lo = chk.clone(), chk.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX); dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        in_sync = bool((hi == lo).item())
        g_ = getattr(model, "_graphed", None)
        comm = {"world_size": world, "backend": dist.get_backend() + (" (RCCL over xGMI), one process per GPU" if dist.get_backend() == "nccl"
                                                                      else " (one-GPU rehearsal of the N > 1 path, not a measurement)"),
                "params_in_sync_after_run": in_sync, "two_segment_backward": bool(g_ is not None and getattr(g_, "bwd_graph2", None) is not None),
                "allreduce_bytes": nbytes,
                "allreduce_ms_alone": ar * 1e3, "bus_GBps": 2.0 * (world - 1) / world * nbytes / ar / 1e9,
                "overlap": "RoI-head FC gradients (58 %) all-reduced under the dense-region backward, RPN head + FPN + DLA "
                           "level5 (27 %) under its second segment, loss vector under backward, the rest after backward"}
        if not in_sync:
            raise RuntimeError("data-parallel ranks hold different parameters after the run")
        del buf
    eager = None
    if world == 1 and mode != "none" and is_main and os.environ.get("CR_BENCH_EAGER", "1") == "1" \
            and not getattr(args, "lean", False):
        # the same step with every kernel launched eagerly (CR_GRAPHS=none): what the dense-region graphs buy
        saved = (model._graphed, model._graphed_cache, model._graphed_max)
        model._graphed, model._graphed_cache, model._graphed_max = None, None, 0
        try:
            n_e = max(1, min(10, args.steps))
            with d2.EventStorage(0):
                for i in range(2):
                    step(batches[i % len(batches)])
                B.barrier(world)
                t1 = time.perf_counter()
                for i in range(n_e):
                    step(batches[i % len(batches)])
                B.barrier(world)
                e_dt = (time.perf_counter() - t1) / n_e
            eager = {"ms_per_step": e_dt * 1e3, "value": IMS_PER_GPU / e_dt, "unit": "images/s", "steps": n_e,
                     "launch_mode": "eager (CR_GRAPHS=none): every kernel of the dense region enqueued by the host"}
        finally:
            model._graphed, model._graphed_cache, model._graphed_max = saved
    ins, ins_err = None, None
    try:                             # (before the micro-benchmark: the profile's last three optimizer updates are these steps)
        with d2.EventStorage(0):
            ins = dominant_kernel_in_step(model, step, batches, dev)
    except Exception as e:           # never lose the headline to the instrumentation
        ins_err = f"{type(e).__name__}: {e}"
    kern = dominant_kernel_roofline(dev, prec)
    try:
        if ins_err:
            raise RuntimeError(ins_err)
        # the roofline figure is the IN-STEP duration (HIP events around the dominant launches inside real train steps, caches
        # in the state the step leaves them in); the back-to-back micro-benchmark stays beside it
        if ins:
            def kname(kind):
                key = {"fwd": "(fwd", "bwd-data": "(bwd-data", "wgrad": "wgrad", "wino-gemm": "k_gemm_batched"}[kind.replace("-group", "")]
                c = [k for k in kern["all_directions"] if key in k]
                return c[0] if c else kind
            worst = min(ins, key=lambda k: ins[k]["tflops"])
            kern["microbench"] = {"kernel": kern["kernel"], "achieved": kern["achieved"], "kernel_ms": kern["kernel_ms"]}
            kern.update(kernel=kname(worst), achieved=ins[worst]["tflops"], frac=ins[worst]["tflops"] / peak,
                        kernel_ms=ins[worst]["ms"], algorithmic_flop_per_launch=ins[worst]["flop_per_launch"],
                        measured="HIP events around the kernel's launches inside train steps (dense region run eagerly for "
                                 "this measurement), mean over %d launches" % ins[worst]["launches"])
            kern["in_step"] = {kname(k): v for k, v in ins.items()}
            kern["traffic"] = pmc_traffic_for(kern["kernel"], prec, IMS_PER_GPU)
    except Exception as e:           # never lose the headline to the instrumentation
        kern["in_step_error"] = f"{type(e).__name__}: {e}"
    note("roofline done")
    import sys
    print(f"[bench] rank {rank} [{prec}]: {ims:.1f} images/s, {dt / args.steps * 1e3:.2f} ms/step", file=sys.stderr, flush=True)
    res = {
        "metric": "images/sec Cube R-CNN DLA34-FPN train step", "value": ims, "unit": "images/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": prec, "data": "synthetic",
        "config": {"workload": "Cube R-CNN DLA34+FPN train step (fwd+loss+bwd+allreduce+SGD), 4 img/GPU 512x512, "
                               "Base_Omni3D.yaml semantics (BASELINE configs[3] per-GPU shard)",
                   "global_batch": IMS_PER_GPU * world, "parallelism": f"dp{world}", "base_lr": cfg.SOLVER.BASE_LR,
                   "precision": PRECISION_TEXT[prec],
                   "launch_mode": {"step": "whole-step HIP graphs (solver.GraphedTrainStep, opt-in)",
                                   "dense": "solver.make_train_step = the step object of do_train / tools/train_net.py: dense region "
                                            "(preprocess, trunk, FPN, RPN head; forward and backward) replayed from per-shape HIP graphs",
                                   "none": "eager"}[mode],
                   "final_loss": rep.get("total_loss"), "skipped_steps": rep.get("iterations_explode"),
                   "valid": bool(rep.get("iterations_explode") == 0 and rep.get("total_loss") == rep.get("total_loss")),
                   "comm": comm},
        "roofline": dict(kern, whole_step={"achieved": achieved_tf, "unit": "TFLOP/s",
                                           "frac": achieved_tf / peak,
                                           "algorithmic_gflop_per_step": TRAIN_GFLOP_PER_IMAGE * IMS_PER_GPU,
                                           "note": "direct-form flops of the step / wall time: an effective rate -- in float32 the "
                                                   "pyramid heads' 3x3 layers execute 2.25x fewer multiplies (Winograd); the "
                                                   "per-kernel figures above count the flops the kernel executes"}),
    }
    if eager is not None:
        res["eager"] = eager
    del step, model, opt
    torch.cuda.empty_cache()
    return res


def bench_do_train(args, rank, world, dev):
    """The product's own loop: `solver.do_train` (what tools/train_net.py calls) for warmup + steps + 1 iterations on host-resident
    uint8 batches fed through `data.DevicePrefetcher` -- pinned staging and the H2D copy of batch i+1 on a side stream under
    step i, LR schedule, the periodic host check of the divergence counters.  The clock runs between the loader handing out
    batch `warmup` and batch `warmup + steps` (device drained at both ends), so it includes the H2D copies the headline leaves out."""
    import shutil
    import tempfile
    syn = importlib.import_module("3dod_amd.synthetic")
    modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
    solver = importlib.import_module("3dod_amd.cubercnn.solver")
    data = importlib.import_module("3dod_amd.cubercnn.data")
    out_dir = tempfile.mkdtemp(prefix="cr3dod_bench_")
    W, K = max(args.warmup, 2), args.steps
    cfg = syn.make_cfg(None, overrides=["MODEL.DEVICE", str(dev), "VIS_PERIOD", 0, "log", False, "SOLVER.IMS_PER_BATCH", 32,
                                        "SOLVER.BASE_LR", 0.02 * IMS_PER_GPU * world / 32.0, "SOLVER.MAX_ITER", W + K + 1,
                                        "SOLVER.CHECKPOINT_PERIOD", 10 ** 9, "TEST.EVAL_PERIOD", 0, "OUTPUT_DIR", out_dir])
    torch.manual_seed(0)
    model = modeling.build_model(cfg)
    host_batches = [syn.make_batch(IMS_PER_GPU, 1234 + rank * 1000 + i) for i in range(4)]      # CPU uint8 images + CPU instances

    def endless():
        i = 0
        while True:
            yield host_batches[i % len(host_batches)]
            i += 1

    class Clock:
        def __init__(self, it):
            self.it, self.n, self.t0, self.t1 = it, 0, None, None

        def __iter__(self):
            return self

        def __next__(self):
            if self.n == W:
                torch.cuda.synchronize(dev)
                self.t0 = time.perf_counter()
            elif self.n == W + K:
                torch.cuda.synchronize(dev)
                self.t1 = time.perf_counter()
            self.n += 1
            return next(self.it)
    clock = Clock(data.DevicePrefetcher(endless(), dev))
    try:
        ok = solver.do_train(cfg, model, clock, world_size=world, rank=rank)
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)
    dt = (clock.t1 - clock.t0) / K
    del model
    torch.cuda.empty_cache()
    return {"ms_per_step": dt * 1e3, "value": IMS_PER_GPU * world / dt, "unit": "images/s", "steps": K, "warmup": W,
            "completed": bool(ok),
            "what": "solver.do_train (tools/train_net.py's loop): DevicePrefetcher feeding host uint8 batches (H2D inside the "
                    "step, overlapped), per-shape dense-region graphs, WarmupMultiStepLR, host check every 20 iterations"}


PYRAMID = ((128, 128), (64, 64), (32, 32), (16, 16), (8, 8))       # p2 ... p6 of a 512 x 512 image


def pyramid_flop(n_img=IMS_PER_GPU, C=256, k=3):
    """2 * M * Cout * k^2 * Cin summed over the five pyramid levels: one grouped launch of the FPN output convolutions or of
    the RPN head's convolution (103.0 GFLOP at 4 images)"""
    return sum(2.0 * n_img * h * w * C * k * k * C for h, w in PYRAMID)


def dominant_kernel_in_step(model, step, batches, dev, n_steps=3):
    """durations of the dominant launches INSIDE train steps: the grouped 3x3 256->256 convolutions over the five pyramid levels
    (FPN output convolutions, RPN head convolution: cr_conv2d_*_group, 103 GFLOP per launch and direction) -- or, where a
    direction is not grouped (bf16 weight gradients), the largest single layer (3x3 256->256 on IMS_PER_GPU x 128 x 128).
    The raw launch wrappers are bracketed with HIP events on the stream they launch on.  The captured dense region is switched
    off for these steps (a graph replay cannot be bracketed per kernel); the kernels, their inputs and the cache state they
    find are those of the real step."""
    ops = importlib.import_module("3dod_amd.hipops")
    shape = (IMS_PER_GPU, 128, 128, 256)
    recs = {"fwd": [], "bwd-data": [], "wgrad": [], "fwd-group": [], "bwd-data-group": [], "wgrad-group": [], "wino-gemm": []}
    T_all = IMS_PER_GPU * sum((h // 2) * (w // 2) for h, w in PYRAMID)
    orig_w = getattr(ops, "wino_gemm_raw", None)

    def wgemm(V, U, M, T, cin, cout):
        f = lambda: orig_w(V, U, M, T, cin, cout)
        return timed("wino-gemm", f) if (T == T_all and cin == 256 and cout == 256) else f()
    orig = (ops.conv_fwd_raw, ops.conv_bwd_data_raw, ops.conv_bwd_weight_raw)
    orig_g = (ops.conv_fwd_group_raw, ops.conv_bwd_data_group_raw, ops.conv_bwd_weight_group_raw)

    def timed(kind, fn, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # keep the queue busy while the launch is prepared: in the faster modes the eager host path does not keep up with
        # the GPU, and an event recorded on an idle queue would count the host's launch latency as kernel time
        torch.cuda._sleep(300000)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        recs[kind].append((e0, e1))
        return r

    def fwd(x, wb, Cout, k, stride, pad, **kw):
        hit = tuple(x.shape) == shape and Cout == 256 and k == 3 and stride == 1
        return timed("fwd", orig[0], x, wb, Cout, k, stride, pad, **kw) if hit else orig[0](x, wb, Cout, k, stride, pad, **kw)

    def bwd(dy, wt, in_shape, k, stride, pad, **kw):
        hit = tuple(in_shape) == shape and dy.shape[3] == 256 and k == 3 and stride == 1
        return timed("bwd-data", orig[1], dy, wt, in_shape, k, stride, pad, **kw) if hit else orig[1](dy, wt, in_shape, k, stride, pad, **kw)

    def wg(dy, x, k, stride, pad, sink=None, bias_acc=None):
        hit = tuple(x.shape) == shape and dy.shape[3] == 256 and k == 3 and stride == 1
        f = lambda: orig[2](dy, x, k, stride, pad, sink, bias_acc)
        return timed("wgrad", lambda: f()) if hit else f()
    is_pyr = lambda ts, k: k == 3 and len(ts) == len(PYRAMID) and tuple(ts[0].shape) == shape

    def gfwd(xs, wbs, ys, Cin, Cout, k, pad, biases, relu):
        f = lambda: orig_g[0](xs, wbs, ys, Cin, Cout, k, pad, biases, relu)
        return timed("fwd-group", f) if is_pyr(xs, k) else f()

    def gbwd(gs, wts, outs, shapes, Cin, Cout, k, pad, accs):
        f = lambda: orig_g[1](gs, wts, outs, shapes, Cin, Cout, k, pad, accs)
        return timed("bwd-data-group", f) if is_pyr(outs, k) else f()

    def gwg(gs, xs, dws, dbs, Cin, Cout, k, pad):
        f = lambda: orig_g[2](gs, xs, dws, dbs, Cin, Cout, k, pad)
        return timed("wgrad-group", f) if is_pyr(xs, k) else f()
    saved = (model._graphed, model._graphed_cache, model._graphed_max)
    model._graphed, model._graphed_cache, model._graphed_max = None, None, 0
    ops.conv_fwd_raw, ops.conv_bwd_data_raw, ops.conv_bwd_weight_raw = fwd, bwd, wg
    ops.conv_fwd_group_raw, ops.conv_bwd_data_group_raw, ops.conv_bwd_weight_group_raw = gfwd, gbwd, gwg
    if orig_w is not None:
        ops.wino_gemm_raw = wgemm
    try:
        for i in range(n_steps):
            step(batches[i % len(batches)])
        torch.cuda.synchronize(dev)
    finally:
        ops.conv_fwd_raw, ops.conv_bwd_data_raw, ops.conv_bwd_weight_raw = orig
        ops.conv_fwd_group_raw, ops.conv_bwd_data_group_raw, ops.conv_bwd_weight_group_raw = orig_g
        if orig_w is not None:
            ops.wino_gemm_raw = orig_w
        model._graphed, model._graphed_cache, model._graphed_max = saved
    flop1, flopg = 2.0 * IMS_PER_GPU * 128 * 128 * 256 * 9 * 256, pyramid_flop()
    out = {}
    for kind, ev in recs.items():
        if ev:
            ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
            fl = 2.0 * 16 * T_all * 256 * 256 if kind == "wino-gemm" else (flopg if kind.endswith("-group") else flop1)
            out[kind] = {"ms": ms, "tflops": fl / ms / 1e9, "launches": len(ev), "flop_per_launch": fl}
    return out


def cpu_baseline_train(inference=False, steps=None, weak=False):
    """the oracle's float32 torch-CPU train step (oracle/cpu_train_step.py) on the box's host cores, in a separate
    process, on a bounded sample (1 warm-up + 4 timed steps of 4 images; inference: 1 + 3 batches of 8 images; weak: 1 + 6
    steps of 2 images of the weakly supervised model)."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    try:
        ncpu = len(os.sched_getaffinity(0))
    except Exception:
        ncpu = os.cpu_count() or 1
    threads = max(1, min(16, ncpu))          # a 1-GPU box gives this job a 16-core share
    cmd = [sys.executable, os.path.join(here, "oracle", "cpu_train_step.py"), "--images", "4", "--steps", str(steps or 4),
           "--warmup", "1", "--threads", str(threads)]
    if weak:
        cmd = cmd[:2] + ["--weak", "--images", "2", "--steps", str(steps or 6), "--warmup", "1", "--threads", str(threads)]
    if inference:
        cmd = cmd[:2] + ["--inference", "--images", "8", "--steps", str(steps or 3), "--warmup", "1", "--threads", str(threads)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads),
               MKL_NUM_THREADS=str(threads))
    print(f"[bench] cpu baseline: {' '.join(cmd[1:])}", file=sys.stderr, flush=True)
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:      # the baseline is a report, never a reason to lose the GPU number
        return {"value": None, "unit": "images/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}


def pmc_traffic_for(kernel, prec, N):
    """memory-side traffic per launch of `kernel` from the committed PMC passes of the same kernel and shapes (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE, separate runs, gfx950 correction applied: profiles/r0N_pmc_*traffic*.json); None if absent"""
    try:
        if N != IMS_PER_GPU:
            raise KeyError("the committed PMC passes are for 4 images")
        here = os.path.dirname(os.path.abspath(__file__))
        fn = {"fp32": "r03_pmc_conv_traffic_fp32.json", "fp32x3": "r02_pmc_conv_traffic_fp32x3.json"}.get(prec, "r01_pmc_conv_traffic.json")
        if kernel.startswith(("k_gemm_batched_f32", "k_wgrad_batched_f32")):
            fn = "r03_pmc_wino_gemm_traffic.json"
        pmc = json.load(open(os.path.join(here, "profiles", fn)))
        key = kernel.split(" ")[0].replace(",", ", ").rstrip(">")
        return [v["traffic_bytes"] for k, v in pmc["kernels"].items() if k.replace(" ", "").startswith(key.replace(" ", ""))][0]
    except Exception:
        return None


def dominant_kernel_roofline(dev, prec="fp32", reps=20, images=IMS_PER_GPU):
    """the dominant launches of the step (profiles/: the implicit-GEMM convolutions) timed live with HIP events on the stream
    they are launched on: the grouped 3x3 256->256 convolution over the five pyramid levels of IMS_PER_GPU 512 x 512 images
    (FPN output convolutions / RPN head convolution; 103.0 GFLOP per launch and direction), warm, back to back; the directions
    that are not grouped in this precision mode are timed on the largest single layer (4 x 128 x 128, 77.3 GFLOP)."""
    ops = importlib.import_module("3dod_amd.hipops")
    dt = torch.float32 if prec != "bf16" else torch.bfloat16
    peak = MFMA_PEAK[prec]
    N, C = images, 256
    g = torch.Generator(device="cpu").manual_seed(0)
    xs = [torch.randn(N, h, w, C, generator=g).to(dev).to(dt) for h, w in PYRAMID]
    dys = [torch.randn(N, h, w, C, generator=g).to(dev).to(dt) for h, w in PYRAMID]
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.02).to(dev).contiguous(memory_format=torch.channels_last)
    wb, wt = ops.prepared_weights(w, True, dt)
    flop1, flopg = 2.0 * N * 128 * 128 * C * 9 * C, pyramid_flop(N, C)
    out = {}
    sink = torch.zeros(C * C * 9, device=dev)           # accumulate target (the flat gradient in the train step)
    grouped = prec != "fp32x3" and ops.group_supported(xs, [w] * len(xs))
    wg1 = {"fp32": "k_conv_wgrad_f32<128,3>", "fp32x3": "k_conv_wgrad_s3_row", "bf16": "k_conv_wgrad<128,3>"}[prec]
    ig1 = {"fp32": "k_conv_igemm_dma<128,3,%d,float>", "fp32x3": "k_conv_igemm_dma_s3<128,3,%d>", "bf16": "k_conv_igemm_dma<128,3,%d>"}[prec]
    tname = "float" if prec == "fp32" else "u16"
    cases = []
    wino = grouped and prec == "fp32" and ops.wino_supported(xs, w, 3, 1)
    if wino:
        # float32 default: forward and backward-data of these layers run the Winograd F(2x2,3x3) route -- their matrix work is
        # the batched GEMM of the 16 transformed positions (RPN head: all five levels in one launch)
        T = sum(x.shape[0] * (x.shape[1] // 2) * (x.shape[2] // 2) for x in xs)
        V = torch.randn(16, T, C, generator=g).to(dev)
        U = (torch.randn(16, C, C, generator=g) * 0.02).to(dev)
        Mo = torch.empty(16, T, C, device=dev)
        lib = importlib.import_module("3dod_amd._lib")

        def gemm16():
            lib.check(lib.load().cr_gemm_batched_f32(lib.ctx_for(V.device), lib.ptr(V), lib.ptr(U), lib.ptr(Mo), T, C, C, 16,
                                                     T * C, C * C, T * C), "cr_gemm_batched_f32")
        cases.append(("k_gemm_batched_f32 (the 16 Winograd products of the RPN head conv, fwd / bwd-data, 5 levels)",
                      2.0 * 16 * T * C * C, gemm16))
    elif grouped:
        ys = [torch.empty_like(x) for x in xs]
        dxs = [torch.empty_like(x) for x in xs]
        cases.append(("k_conv_igemm_dma_grp<3,0,%s> (fwd, 5 levels)" % tname, flopg,
                      lambda: ops.conv_fwd_group_raw(xs, [wb] * 5, ys, C, C, 3, 1, [None] * 5, False)))
        cases.append(("k_conv_igemm_dma_grp<3,1,%s> (bwd-data, 5 levels)" % tname, flopg,
                      lambda: ops.conv_bwd_data_group_raw(dys, [wt] * 5, dxs, [x.shape for x in xs], C, C, 3, 1, [None] * 5)))
    else:
        cases.append((ig1 % 0 + " (fwd)", flop1, lambda: ops.conv_fwd_raw(xs[0], wb, C, 3, 1, 1)))
        cases.append((ig1 % 1 + " (bwd-data)", flop1, lambda: ops.conv_bwd_data_raw(dys[0], wt, xs[0].shape, 3, 1, 1)))
    if wino and ops.wino_wgrad_on():
        # ... and so does their weight gradient: dU[k] = dM[k]^T V[k] for the 16 positions in one launch (f32 atomics over 12
        # pixel splits into a zeroed dU; the zero-fill is part of what is timed)
        dM = torch.randn(16, T, C, generator=g).to(dev)
        dU = torch.empty(16, C, C, device=dev)

        def wgrad16():
            dU.zero_()
            lib.check(lib.load().cr_wgrad_batched_f32(lib.ctx_for(V.device), lib.ptr(dM), lib.ptr(V), lib.ptr(dU), T, C, C, 16,
                                                      T * C, T * C, C * C), "cr_wgrad_batched_f32")
        cases.append(("k_wgrad_batched_f32 (the 16 Winograd weight-gradient products of the RPN head conv, 5 levels)",
                      2.0 * 16 * T * C * C, wgrad16))
    elif grouped and prec == "fp32":
        cases.append(("k_conv_wgrad_f32_grp<3> (wgrad, 5 levels)", flopg,
                      lambda: ops.conv_bwd_weight_group_raw(dys, xs, [sink] * 5, [None] * 5, C, C, 3, 1)))
    else:
        cases.append((wg1, flop1, lambda: ops.conv_bwd_weight_raw(dys[0], xs[0], 3, 1, 1, sink=sink)))
    for name, flop, fn in cases:
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / reps
        out[name] = {"ms": ms, "tflops": flop / ms / 1e9, "flop_per_launch": flop}
    worst = min(out, key=lambda k: out[k]["tflops"])
    # memory-side traffic per launch from the committed PMC passes of the same kernel and shapes (rocprofv3 --pmc
    # FETCH_SIZE / WRITE_SIZE, separate runs, gfx950 correction applied: profiles/r0N_pmc_conv_traffic*.json)
    traffic = pmc_traffic_for(worst, prec, N)
    return {"bound": "mfma", "kernel": worst,
            "shape": ((f"3x3 conv 256->256 on the five pyramid levels of {N} x 512 x 512 images ({N}x128x128 ... {N}x8x8; FPN output convs / "
                       "RPN head conv), one grouped launch, ") if "levels" in worst else f"3x3 conv 256->256 on {N}x128x128 (FPN p2 output), ")
                     + ("Winograd F(2x2,3x3): 16 GEMMs (tiles x 256) @ (256 x 256) in one launch, " if worst.startswith("k_gemm_batched") else "")
                     + {"fp32": "f32 in / f32 acc", "fp32x3": "f32 in (3 x bf16 split, 6 MFMAs per term) / f32 acc", "bf16": "bf16 in / f32 acc"}[prec],
            "achieved": out[worst]["tflops"], "peak": peak, "unit": "TFLOP/s",
            "frac": out[worst]["tflops"] / peak, "traffic": traffic,
            "algorithmic_flop_per_launch": out[worst]["flop_per_launch"], "kernel_ms": out[worst]["ms"], "all_directions": out}
