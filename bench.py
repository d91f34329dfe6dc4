#!/usr/bin/env python
"""bench.py -- measures the hot path on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload train|geometry|inference|weak|depth]

N>1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

Workloads
  geometry  BASELINE.json configs[2]: ProposalNetwork 1000-cube/object project + IoU/size/corner
            scoring + argmax, 64 images x 16 objects x 1000 cubes per GPU, ONE launch per step.
  train     BASELINE.json metric: Cube R-CNN DLA34-FPN train step (see bench_train.py).
  inference BASELINE.json configs[1] (secondary): the same model in eval mode, 8 images per GPU.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def dist_setup(n_gpus):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # CR_REHEARSE_ONE_GPU=1: every rank on cuda:0 with gloo collectives on the device tensors -- a functional rehearsal of the
    # N > 1 code path (two-segment backward, three all-reduce phases, broadcast) on a one-GPU box; not a measurement
    rehearse = os.environ.get("CR_REHEARSE_ONE_GPU") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    return rank, world, local


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world, dev):
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---------------------------------------------------------------------------
def geometry_inputs(n_obj, P, seed, dev):
    """Synthetic Omni3D-shaped scoring inputs: K f~U(400,800), pp (256,256);
    cubes around plausible depths; 2D boxes; priors mu~U(0.3,1.1), sigma=0.2mu; a rect per object."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    f = float(torch.empty(1).uniform_(400, 800, generator=g))
    K = torch.tensor([[f, 0, 256], [0, f, 256], [0, 0, 1]], dtype=torch.float32)
    z = torch.empty(n_obj, P).uniform_(1, 8, generator=g)
    u = torch.empty(n_obj, P).uniform_(0, 512, generator=g)
    v = torch.empty(n_obj, P).uniform_(0, 512, generator=g)
    x = (u - 256) * z / f
    y = (v - 256) * z / f
    dims = torch.empty(n_obj, P, 3).uniform_(0.05, 1.5, generator=g)
    yaw = torch.empty(n_obj, P).uniform_(0, 3.14159, generator=g)
    c, s = yaw.cos(), yaw.sin()
    zeros, ones = torch.zeros_like(c), torch.ones_like(c)
    R = torch.stack([c, zeros, s, zeros, ones, zeros, -s, zeros, c], -1)
    cubes = torch.cat([x[..., None], y[..., None], z[..., None], dims, R], -1).float()
    ctr = torch.empty(n_obj, 2).uniform_(64, 448, generator=g)
    wh = torch.empty(n_obj, 2).uniform_(32, 256, generator=g)
    ref = torch.cat([ctr - wh / 2, ctr + wh / 2], 1).clamp(0, 511)
    mu = torch.empty(n_obj, 3).uniform_(0.3, 1.1, generator=g)
    sg = 0.2 * mu
    rect = torch.stack([ref[:, [0, 1]], ref[:, [2, 1]], ref[:, [2, 3]], ref[:, [0, 3]]], 1).contiguous()
    to = lambda t: t.to(dev).contiguous()
    return dict(cubes=to(cubes), K=to(K), im_wh=(512, 512), ref=to(ref), mu=to(mu), sg=to(sg), rect=to(rect))


def pmc_traffic(fname, kernel_prefix):
    """HBM bytes per launch of a kernel from the committed PMC passes (profiles/, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate runs, gfx950 correction applied by scripts/pmc_summarize.py); None when the file is absent"""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", fname)))
        return [v["traffic_bytes"] for k, v in d["kernels"].items() if k.startswith(kernel_prefix)][0]
    except Exception:
        return None


def bench_geometry(args, rank, world, dev, argmax_only=False, inp=None, cpu_baseline=True, exact=False):
    """argmax_only: the AP path of roi_heads.py:501-505 -- only the best cube's index and score leave the kernel
    (60 B/cube read, O(1) per object written); default: corners, boxes and the four score planes are written too (156 B/cube).
    exact=False: cr_cubes_project_score_fast (argmax and best score bit-equal to the oracle, planes to 1e-4: north_star's
    contract); exact=True: cr_cubes_project_score (every plane bit-equal to the oracle)."""
    geo = importlib.import_module("3dod_amd.geometry")
    n_img, n_obj_img, P = 64, 16, 1000
    n_obj = n_img * n_obj_img
    if inp is None:
        inp = geometry_inputs(n_obj, P, 1234 + rank, dev)
    want = () if argmax_only else ("corners", "boxes", "iou", "dim", "corner", "combined")

    # the output planes are allocated once and written again by every launch (`out=`): eight allocations per call are ~37 us
    # of host time, more than the fast kernel takes -- the loop would measure the host (scripts/geo_alloc_probe.py)
    bufs = geo.cubes_project_score(inp["cubes"], inp["K"], inp["im_wh"], inp["ref"], inp["mu"], inp["sg"], inp["rect"],
                                   want=want, fast=not exact)

    def step():
        return geo.cubes_project_score(inp["cubes"], inp["K"], inp["im_wh"], inp["ref"], inp["mu"], inp["sg"],
                                       inp["rect"], want=want, fast=not exact, out=bufs)
    for _ in range(args.warmup):
        step()
    # the K launches are one HIP graph (CR_GRAPHS=none: K host launches): the wrapper's host time per call (~35 us of argument
    # checks and ctypes marshalling) is about what the fast kernel takes, so a host loop would time whichever of the two is slower
    graph = None
    if os.environ.get("CR_GRAPHS", "dense") != "none":
        try:
            torch.cuda.synchronize()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    for _ in range(args.steps):
                        step()
            torch.cuda.current_stream().wait_stream(side)
            graph.replay()                                  # one untimed replay
            torch.cuda.synchronize()
        except Exception as e:                              # (a capture problem must not cost the measurement)
            print(f"[bench] geometry: graph capture failed ({type(e).__name__}: {e}); timing host launches", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    barrier(world)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    if graph is not None:
        graph.replay()
    else:
        for _ in range(args.steps):
            step()
    ev1.record()
    barrier(world)
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dt, world, dev)
    kern_ms = ev0.elapsed_time(ev1) / args.steps       # events on the stream the kernel is launched on
    cubes_total = n_obj * P * world
    value = cubes_total * args.steps / dt
    bytes_per_cube = 60 if argmax_only else 60 + 64 + 16 + 16    # read 60 B (+ corners 64 + box 16 + 4 score planes)
    achieved = bytes_per_cube * n_obj * P / (kern_ms * 1e-3) / 1e9
    res = {
        "metric": "cubes/sec ProposalNetwork 1000-cube project+score+argmax (BASELINE configs[2])"
                  + (", argmax-only outputs" if argmax_only else "") + (", bit-exact planes" if exact else ""),
        "value": value, "unit": "cubes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "geometry: 64 images x 16 objects x 1000 cubes per GPU, "
                               + ("argmax + best score only (60 B/cube)" if argmax_only else "full outputs (156 B/cube)") + ", one launch; "
                               + ("every plane bit-equal to the oracle" if exact else
                                  "argmax / best score bit-equal to the oracle, planes to 1e-4 (cr_cubes_project_score_fast)"),
                   "objects_per_gpu": n_obj, "proposals": P, "parallelism": f"objects sharded x{world}, no collective",
                   "launch_mode": "the K launches replayed as one HIP graph" if graph is not None else "K host launches"},
        "roofline": {"bound": "hbm", "kernel": GEOMETRY_KERNEL[exact] + (" (no output planes)" if argmax_only else ""),
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": pmc_traffic(GEOMETRY_PMC[(exact, argmax_only)], "k_project_score"),
                     "algorithmic_bytes_per_launch": bytes_per_cube * n_obj * P, "kernel_ms": kern_ms},
    }
    if rank == 0 and world == 1 and cpu_baseline and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_geometry(inp, P)
    return res


GEOMETRY_KERNEL = {False: "k_project_score_fast<4>", True: "k_project_score<4>"}
GEOMETRY_PMC = {(True, False): "r02_pmc_geometry_traffic.json", (True, True): "r03_pmc_geometry_argmax_traffic.json",
                (False, False): "r03_pmc_geometry_fast_traffic.json", (False, True): "r03_pmc_geometry_fast_argmax_traffic.json"}


def cpu_baseline_geometry(inp, P):
    """oracle (numpy restatement of the reference's per-object loop, roi_heads.py:494-505) on a bounded sample."""
    from oracle import geometry as og
    n = 1024
    c = {k: (v[:n].cpu().numpy() if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] > 3 else v) for k, v in inp.items()}
    K = inp["K"].cpu().numpy()
    og.project_and_score(c["cubes"][:2], K, inp["im_wh"], c["ref"][:2], c["mu"][:2], c["sg"][:2], c["rect"][:2])
    t0 = time.perf_counter()
    og.project_and_score(c["cubes"], K, inp["im_wh"], c["ref"], c["mu"], c["sg"], c["rect"])
    dt = time.perf_counter() - t0
    return {"value": n * P / dt, "unit": "cubes/s", "cores": 1, "kind": "port",
            "sample": f"{n} objects x {P} cubes, numpy float32 oracle, per-object loop, {dt:.2f} s"}


def bench_inference(args, rank, world, dev):
    """BASELINE.json configs[1] (secondary metric): Cube R-CNN DLA34-FPN inference, 8 synthetic 512x512 images per GPU,
    random-init weights, full post-processing (NMS, top-100, cube decoding) -> images/s."""
    bt = importlib.import_module("bench_train")
    d2 = importlib.import_module("3dod_amd.d2lite")
    cfg, model, opt, syn, solver = bt.build(dev, world=world)
    model.eval()
    B = 8
    batches = [syn.make_batch(B, 4321 + rank * 100 + i, with_gt=False) for i in range(2)]
    for b in batches:
        for d in b:
            d["image"] = d["image"].to(dev)
    if os.environ.get("CR_GRAPHS", "dense") != "none":
        with torch.no_grad(), d2.EventStorage(0):
            model.enable_graphs_eval(batches[0])         # forward-only HIP graph of preprocess + trunk + FPN + RPN head
    n_det = 0
    with torch.no_grad(), d2.EventStorage(0):
        for i in range(args.warmup):
            model(batches[i % 2])
        barrier(world)
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = model(batches[i % 2])
            n_det += sum(len(o["instances"]) for o in out)
        barrier(world)
        dt = max_over_ranks(time.perf_counter() - t0, world, dev)
    prec = importlib.import_module("3dod_amd.hipops").precision()
    gflop_img = 116.8                    # 58.4 GMAC per 512x512 image with 1000 RoIs
    ach = gflop_img * B / (dt / args.steps) / 1e3
    res = _inference_result(args, world, B, dt, n_det, prec, ach, bt.MFMA_PEAK[prec], gflop_img)
    del model, opt
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = bt.cpu_baseline_train(inference=True, steps=getattr(args, "cpu_steps", None))
    return res


def _inference_result(args, world, B, dt, n_det, prec, ach, peak, gflop_img):
    return {"metric": "images/sec Cube R-CNN DLA34-FPN inference (BASELINE configs[1])", "value": B * world * args.steps / dt,
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": prec, "data": "synthetic",
            "config": {"workload": "Cube R-CNN DLA34+FPN inference, 8 img/GPU 512x512, random-init weights, full post-processing",
                       "global_batch": B * world, "parallelism": f"dp{world}", "detections_per_step": n_det / max(args.steps, 1)},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                         "scope": "whole forward incl. post-processing (algorithmic flops / wall time); per-kernel figures: "
                                  "the train line's roofline (same conv kernels)", "algorithmic_gflop_per_image": gflop_img}}


def bench_boxnet(args, rank, world, dev):
    """BASELINE.json configs[2] end to end: BoxNet on GT boxes, 64 synthetic 512x512 images x 16 objects per GPU with depth,
    ground and object masks resident in HBM -> images/s through ground-plane RANSAC, 1000 proposals per object, mask ->
    minimum-area rectangle, project + score + argmax and the Instances packing (`--workload geometry` times the scoring
    kernel alone)."""
    syn = importlib.import_module("3dod_amd.synthetic")
    modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
    here = os.path.dirname(os.path.abspath(__file__))
    cfg = syn.make_cfg(os.path.join(here, "configs", "BoxNet.yaml"), ["MODEL.DEVICE", str(dev), "VIS_PERIOD", 0, "log", False])
    torch.manual_seed(0)
    model = modeling.build_model(cfg).eval()
    B, n_obj = 64, 16
    batch = syn.make_batch(B, 777 + rank, min_obj=n_obj, max_obj=n_obj)
    g = torch.Generator().manual_seed(2 + rank)
    yy = torch.arange(512)[:, None]
    for b in batch:
        b["image"] = b["image"].to(dev)
        b["instances"] = b["instances"].to(dev)
        b["depth_map"] = (torch.rand(512, 512, generator=g) * 3 + 1).to(dev)
        b["ground_map"] = (yy > 300).expand(512, 512).to(torch.uint8).to(dev)
        boxes = b["instances"].gt_boxes.tensor.round().long().clamp(0, 511).cpu()
        m = torch.zeros(len(boxes), 512, 512, dtype=torch.bool)
        xx = torch.arange(512)[None, :]
        for j, bb in enumerate(boxes):                     # an ellipse inscribed in the box: a non-trivial hull
            cx, cy, rx, ry = (bb[0] + bb[2]) / 2, (bb[1] + bb[3]) / 2, (bb[2] - bb[0]) / 2 + 0.5, (bb[3] - bb[1]) / 2 + 0.5
            m[j] = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1
        b["masks"] = m.to(dev)
    nobj = sum(len(b["instances"]) for b in batch)
    gen = torch.Generator(device=dev).manual_seed(3 + rank)
    run = lambda: model.inference(batch, experiment_type={"use_pred_boxes": False}, generator=gen)
    with torch.no_grad():
        for _ in range(args.warmup):
            run()
        barrier(world)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        barrier(world)
        dt = max_over_ranks(time.perf_counter() - t0, world, dev)
    # roofline of the pipeline's dominant stage (rocprofv3: k_ccl_merge / k_mask_rect / k_ccl_sizes lead the kernel time):
    # the six kernels of cr_mask_rects on this batch's masks, HIP events on the launch stream.  Algorithmic bytes (DESIGN
    # section 2): the mask once (1 B / pixel of every 512 x 512 mask: the bounding-window search reads all of it) + one
    # 4-B label written and read once per pixel of each mask's bounding window.
    geo = importlib.import_module("3dod_amd.geometry")
    masks = [b["masks"] for b in batch]
    for _ in range(3):
        geo.mask_rects(masks)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10):
        geo.mask_rects(masks)
    ev1.record()
    torch.cuda.synchronize()
    stage_ms = ev0.elapsed_time(ev1) / 10
    win_px = 0
    for b in batch:
        bb = b["instances"].gt_boxes.tensor.round().long().clamp(0, 511).cpu()
        win_px += int(((bb[:, 2] - bb[:, 0] + 1) * (bb[:, 3] - bb[:, 1] + 1)).sum())
    alg_bytes = nobj * 512 * 512 + 8 * win_px
    achieved = alg_bytes / (stage_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": "cr_mask_rects = k_mask_bbox + k_ccl_init/merge/sizes/best + k_mask_rect (mask -> largest "
                                          "component -> hull -> minimum-area rectangle)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": stage_ms, "note": "union-find label passes are latency-bound (atomicMin parent chains), not "
                                               "bandwidth-bound; share of the step = kernel_ms / ms_per_step"}
    extra = {"roofline": roofline}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        extra["cpu_baseline"] = cpu_baseline_boxnet(batch[0], n_obj, seconds=getattr(args, "cpu_seconds", 12.0))
    return {**extra,
            "metric": "images/sec BoxNet 1000-cube proposal-and-scoring pipeline on GT boxes (BASELINE configs[2], end to end)",
            "value": B * world * args.steps / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BoxNet.inference(use_pred_boxes=False): {B} images x {nobj // B} objects x 1000 cubes per GPU, "
                                   "depth + ground + object masks in HBM", "objects_per_gpu": nobj,
                       "cubes_per_s": nobj * 1000 * world * args.steps / dt, "parallelism": f"images sharded x{world}, no collective"}}


def cpu_baseline_boxnet(sample, n_obj, P=1000, seconds=12.0):
    """the oracle's stages of the same pipeline on ONE image (16 objects x 1000 cubes), numpy on one host core: ground-plane
    RANSAC on the back-projected ground pixels (oracle.geometry.ransac_plane), mask -> minimum-area rectangle
    (oracle.rect.rect_from_mask, scipy labelling), proposals from supplied draws (propose_from_draws), project + score +
    argmax (project_and_score)."""
    import numpy as np
    from oracle import geometry as og, rect as orect
    rng = np.random.default_rng(0)
    depth = sample["depth_map"].cpu().numpy().astype(np.float32)
    ground = sample["ground_map"].cpu().numpy().astype(bool)
    masks = sample["masks"].cpu().numpy()
    boxes = sample["instances"].gt_boxes.tensor.cpu().numpy().astype(np.float32)[:n_obj]
    f = 600.0
    K = np.array([[f, 0, 256], [0, f, 256], [0, 0, 1]], dtype=np.float32)
    mu = rng.uniform(0.3, 1.1, (n_obj, 3)).astype(np.float32)
    sg = (0.2 * mu).astype(np.float32)
    n_rep = 0                                                        # whole images for ~`seconds` of host work
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        n_rep += 1
        ys, xs = np.nonzero(ground)
        sel = rng.choice(len(ys), min(len(ys), 20000), replace=False)     # bounded: the 1000 x points distance matrix is host RAM
        ys, xs = ys[sel], xs[sel]
        z = depth[ys, xs]
        pts = np.stack([(xs - 256) * z / f, (ys - 256) * z / f, z], 1).astype(np.float32)
        triples = rng.integers(0, len(pts), (1000, 3))
        eq = og.ransac_plane(pts, triples, 0.05)[0]
        normal = og.fix_ground_normal(-eq[:3] / (np.linalg.norm(eq[:3]) + 1e-12))
        rects = []
        for m, b in zip(masks[:n_obj], boxes):
            r = orect.rect_from_mask(m)
            rects.append(r if r is not None else np.array([[b[0], b[1]], [b[2], b[1]], [b[2], b[3]], [b[0], b[3]]], np.float32))
        cubes = og.propose_from_draws(boxes.clip(0, 511), depth, mu, sg, K, P, rng.standard_normal((4, 3, n_obj, P)).astype(np.float32),
                                      rng.standard_normal((3, n_obj, P)).astype(np.float32), rng.integers(0, 36, (n_obj, P)), normal)
        og.project_and_score(cubes, K, (512, 512), boxes, mu, sg, np.stack(rects).astype(np.float32))
    dt = time.perf_counter() - t0
    return {"value": n_rep / dt, "unit": "images/s", "cores": 1, "kind": "port",
            "sample": f"{n_rep} x 1 image x {n_obj} objects x {P} cubes through the oracle's stages (numpy / scipy, float32; RANSAC on 20000 of the "
                      f"ground pixels), {dt:.2f} s"}


def bench_weak(args, rank, world, dev):
    """BASELINE.json configs[4] without its Depth-Anything backbone (not built): the weakly supervised Cube R-CNN
    (configs/Omni_combined.yaml: RCNN3D_combined_features + ROIHeads3DScore, losses from 2D boxes + depth / ground maps),
    2 synthetic 512x512 images per GPU (16 on 8 GPUs) with precomputed depth and ground maps -> images/s of the train step."""
    bt = importlib.import_module("bench_train")
    d2 = importlib.import_module("3dod_amd.d2lite")
    B = 2
    cfg, model, opt, syn, solver = bt.build(dev, world=world, config="Omni_combined.yaml", lr=0.015 * B * world / 25.0)
    if world > 1:
        import torch.distributed as dist
        dist.broadcast(opt.flat_p, 0)
    batches = [syn.add_scene_maps(syn.make_batch(B, 777 + rank * 1000 + i), 99 + i, ground_every=2) for i in range(4)]
    for b in batches:
        for d in b:
            d["image"], d["instances"], d["depth_map"] = d["image"].to(dev), d["instances"].to(dev), d["depth_map"].to(dev)
            if d["ground_map"] is not None:
                d["ground_map"] = d["ground_map"].to(dev)
    step = solver.TrainStep(cfg, model, opt, world_size=world)
    if os.environ.get("CR_GRAPHS", "dense") != "none":
        model.enable_graphs(batches[0])
        opt.zero_grad()
    live = os.environ.get("CR_LIVE_DEPTH", "0") == "1"
    if live:
        # the multi-model pipeline of configs[4]: the depth maps are not precomputed but come from the Depth-Anything-V2
        # ViT-L forward on the same images inside the timed step (resize to 518, ImageNet normalisation, resize back)
        dav2 = importlib.import_module("3dod_amd.depth_anything_v2")
        depth_model = dav2.DepthAnythingV2("vitl")
        depth_model.load_state_dict(syn.seeded_state_dict(depth_model, 0))
        depth_model = depth_model.to(dev).eval()
        mean = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1)
        std = torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1)

        def with_live_depth(batch):
            import torch.nn.functional as F
            with torch.no_grad():
                rgb = torch.stack([d["image"] for d in batch]).flip(1).float() / 255.0              # BGR uint8 -> RGB [0,1]
                x = F.interpolate((rgb - mean) / std, (518, 518), mode="bilinear", align_corners=False)
                depth = depth_model(x)
                depth = F.interpolate(depth[:, None], rgb.shape[-2:], mode="bilinear", align_corners=True)[:, 0]
            return [dict(d, depth_map=depth[i]) for i, d in enumerate(batch)]
    else:
        with_live_depth = lambda batch: batch
    with d2.EventStorage(1):
        for i in range(args.warmup):
            step(with_live_depth(batches[i % len(batches)]))
        barrier(world)
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(with_live_depth(batches[i % len(batches)]))
        barrier(world)
        dt = max_over_ranks(time.perf_counter() - t0, world, dev)
        rep = step.report()
    extra = {}
    if rank == 0:
        # the dominant launches of this step are the same grouped pyramid convolutions as in the supervised step, at 2 images
        extra["roofline"] = bt.dominant_kernel_roofline(dev, importlib.import_module("3dod_amd.hipops").precision(), images=B)
        if world == 1 and not args.no_cpu_baseline:
            extra["cpu_baseline"] = bt.cpu_baseline_train(weak=True)
    return {**_weak_line(args, world, dt, rep, live, cfg, B), **extra}


def _weak_line(args, world, dt, rep, live, cfg, B):
    return {"metric": "images/sec weakly supervised Cube R-CNN train step (BASELINE configs[4], "
                      + ("live Depth-Anything-V2 depth maps)" if live else "precomputed depth maps)"),
            "value": B * world * args.steps / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": importlib.import_module("3dod_amd.hipops").precision(), "data": "synthetic",
            "config": {"workload": "RCNN3D_combined_features + ROIHeads3DScore train step, 2 img/GPU 512x512, losses "
                                   + ",".join(cfg.loss_functions) + ("; depth maps from the Depth-Anything-V2 ViT-L forward "
                                   "inside the step (CR_LIVE_DEPTH=1)" if live else "; precomputed depth maps (as the reference trains)"),
                       "global_batch": B * world, "parallelism": f"dp{world}", "final_loss": rep.get("total_loss"),
                       "skipped_steps": rep.get("iterations_explode"),
                       "valid": bool(rep.get("iterations_explode") == 0 and rep.get("total_loss") == rep.get("total_loss"))}}


def depth_model_gflop(N_img, depth, D, heads_out, features, ph, pw):
    """algorithmic GFLOP of one Depth-Anything-V2 forward per image: ViT blocks (24 N D^2 + 4 N^2 D per layer), patch
    embedding, DPT head convolutions / deconvolutions (2 * MACs)."""
    N = ph * pw + 1
    f = depth * (24.0 * N * D * D + 4.0 * N * N * D) + 2.0 * ph * pw * 588 * D
    oc = heads_out
    res = [(ph * 4, pw * 4), (ph * 2, pw * 2), (ph, pw), ((ph + 1) // 2, (pw + 1) // 2)]
    for i in range(4):
        f += 2.0 * ph * pw * D * oc[i]                                      # 1x1 projects
        f += 2.0 * res[i][0] * res[i][1] * oc[i] * 9 * features             # layerN_rn 3x3
    f += 2.0 * ph * pw * oc[0] * oc[0] * 16 + 2.0 * ph * pw * oc[1] * oc[1] * 4 + 2.0 * res[3][0] * res[3][1] * oc[3] * oc[3] * 9
    rcu = lambda hw: 2 * 2.0 * hw[0] * hw[1] * features * features * 9
    f += rcu(res[3]) + 2 * rcu(res[2]) + 2 * rcu(res[1]) + 2 * rcu(res[0])   # refinenet4 has one unit, the others two
    for hw in (res[2], res[1], res[0], (ph * 8, pw * 8)):
        f += 2.0 * hw[0] * hw[1] * features * features                       # out_conv 1x1 after each up-sampling
    f += 2.0 * (ph * 8) * (pw * 8) * features * 9 * (features // 2)
    f += 2.0 * (ph * 14) * (pw * 14) * ((features // 2) * 9 * 32 + 32)
    return f / 1e9


def bench_depth(args, rank, world, dev):
    """The depth backbone of BASELINE.json configs[4] on its own: Depth-Anything-V2 ViT-L + DPT head, 518x518 inputs,
    random-init weights (no checkpoint offline), 4 images per GPU per step -> images/s and the achieved fraction of the
    bf16 MFMA peak over the whole forward (GEMMs are hipBLASLt, attention / convolutions / norms are this library's)."""
    dav2 = importlib.import_module("3dod_amd.depth_anything_v2")
    syn = importlib.import_module("3dod_amd.synthetic")
    B, S = 4, 518
    torch.manual_seed(0)
    model = dav2.DepthAnythingV2("vitl")
    model.load_state_dict(syn.seeded_state_dict(model, 0))
    model = model.to(dev).eval()
    xs = [torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(10 + rank * 7 + i)).to(dev) for i in range(2)]
    # ~330 launches per forward with 12 ms of kernels: the host is the bottleneck when they are enqueued one by one (18 ms per
    # step), so the forward is captured once as a HIP graph over a static input buffer (CR_GRAPHS=none: eager)
    graphed = os.environ.get("CR_GRAPHS", "dense") != "none"
    with torch.no_grad():
        for i in range(max(args.warmup, 2)):
            model(xs[i % 2])
        if graphed:
            static_x = xs[0].clone()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_d = model(static_x)

            def run(x):
                static_x.copy_(x)
                graph.replay()
                return static_d
        else:
            run = model
        run(xs[1])
        barrier(world)
        t0 = time.perf_counter()
        for i in range(args.steps):
            d = run(xs[i % 2])
        barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, dev)
    assert bool(torch.isfinite(d).all())
    gflop = depth_model_gflop(B, 24, 1024, [256, 512, 1024, 1024], 256, S // 14, S // 14)
    ach = gflop * B / (dt / args.steps) / 1e3
    extra = {}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        extra["cpu_baseline"] = cpu_baseline_depth(model, S)
    return {**extra, "metric": "images/sec Depth-Anything-V2 ViT-L forward (depth backbone of BASELINE configs[4])",
            "value": B * world * args.steps / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "Depth-Anything-V2 ViT-L + DPT head forward, 4 img/GPU 518x518 (1370 tokens), seeded random weights"
                                   + (", forward replayed as one HIP graph" if graphed else ", eager launches"),
                       "global_batch": B * world, "parallelism": f"dp{world}", "gflop_per_image": gflop},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0,
                         "traffic": None, "scope": "whole forward (algorithmic flops / wall time)"}}


def cpu_baseline_depth(model, S, seconds=15.0):
    """oracle/depth_ref.py (plain torch float32 restatement of the reference's forward, pinned against the reference's own
    output by tests/test_depth_oracle.py) with the SAME weights on the host cores: whole 518 x 518 images for ~15 s"""
    from oracle import depth_ref
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    x = torch.randn(1, 3, S, S, generator=torch.Generator().manual_seed(99))
    with torch.no_grad():
        depth_ref.forward(sd, x, "vitl", model.max_depth)                  # warm-up (thread pool, allocator)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            depth_ref.forward(sd, x, "vitl", model.max_depth)
            n += 1
        dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n} x 1 image 518 x 518 through oracle/depth_ref.py (torch float32, {cores} threads), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="train", choices=["train", "geometry", "inference", "weak", "depth", "boxnet"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--argmax-only", action="store_true", help="geometry workload: the AP path (no output planes)")
    ap.add_argument("--exact-planes", action="store_true", help="geometry workload: cr_cubes_project_score (bit-exact planes)")
    ap.add_argument("--train-only", action="store_true", help="train workload without the geometry / inference keys")
    ap.add_argument("--lean", action="store_true", help="headline measurement only (no eager / do_train / other-precision lines)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 200 if args.workload == "geometry" else 20
    if args.warmup is None:
        args.warmup = 20 if args.workload == "geometry" else 5
    rank, world, local = dist_setup(args.gpus)
    dev = torch.device("cuda", local)
    if args.workload == "geometry":
        res = bench_geometry(args, rank, world, dev, argmax_only=args.argmax_only, cpu_baseline=not args.argmax_only,
                             exact=args.exact_planes)
    elif args.workload == "inference":
        res = bench_inference(args, rank, world, dev)
    elif args.workload == "weak":
        res = bench_weak(args, rank, world, dev)
    elif args.workload == "depth":
        res = bench_depth(args, rank, world, dev)
    elif args.workload == "boxnet":
        res = bench_boxnet(args, rank, world, dev)
    else:
        bt = importlib.import_module("bench_train")
        res = bt.bench_train(args, rank, world, dev)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = bt.cpu_baseline_train(steps=2)
        if world == 1 and not args.train_only:
            # the other two workloads north_star names, in the SAME line (labelled keys; each with its own roofline and
            # cpu_baseline): the 1000-cube geometry (BASELINE configs[2]; full outputs, the argmax-only AP path, the whole
            # BoxNet proposal-and-scoring pipeline around it) and detector inference at 8 x 512 x 512 (configs[1]).
            # `value` above stays the train step.
            sub = argparse.Namespace(**vars(args))
            keep = ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "roofline", "cpu_baseline")
            for key, fn, st, wu, kw in (("geometry", bench_geometry, 200, 20, {}),
                                        ("geometry_argmax_only", bench_geometry, 200, 20, {"argmax_only": True, "cpu_baseline": False}),
                                        ("geometry_exact_planes", bench_geometry, 200, 20, {"exact": True, "cpu_baseline": False}),
                                        ("boxnet", bench_boxnet, 20, 5, {}),
                                        ("inference", bench_inference, 20, 5, {})):
                sub.steps, sub.warmup, sub.cpu_steps, sub.cpu_seconds = st, wu, 1, 4.0
                try:
                    r = fn(sub, rank, world, dev, **kw)
                    res[key] = {k: r[k] for k in keep if k in r}
                except Exception as e:            # never lose the headline to a secondary workload
                    res[key] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
