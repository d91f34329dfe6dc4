"""ORACLE / CPU baseline (test infrastructure): runs the Cube R-CNN DLA34-FPN train step on the host cores in
plain PyTorch float32 (oracle/cpu_backend.py under the product's host-side model) and prints one JSON line.

    python oracle/cpu_train_step.py --images 2 --steps 2 --warmup 1 [--threads N]
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=2)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--inference", action="store_true", help="time model.inference (eval mode) instead of the train step")
    ap.add_argument("--weak", action="store_true", help="the weakly supervised model of configs/Omni_combined.yaml (depth / ground maps)")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    from oracle import cpu_backend
    cpu_backend.install()
    syn = importlib.import_module("3dod_amd.synthetic")
    modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
    solver = importlib.import_module("3dod_amd.cubercnn.solver")
    d2 = importlib.import_module("3dod_amd.d2lite")
    cfg_file = os.path.join(ROOT, "configs", "Omni_combined.yaml") if args.weak else None
    cfg = syn.make_cfg(cfg_file, overrides=["MODEL.DEVICE", "cpu", "VIS_PERIOD", 0, "log", False, "SOLVER.BASE_LR", 0.02])
    torch.manual_seed(0)
    model = cpu_backend.attach(modeling.build_model(cfg))
    if args.weak:
        # the two kernels of the tensor composition (window median, RANSAC plane) as their CPU restatements
        from oracle import weak as ow
        model.roi_heads._median_fn, model.roi_heads._plane_cls = ow.box_median, ow.Plane
    if args.inference:
        model.eval()
        batches = [syn.make_batch(args.images, 4321 + i, size=args.size, with_gt=False) for i in range(2)]
        with torch.no_grad(), d2.EventStorage(0):
            for i in range(args.warmup):
                model(batches[i % 2])
            t0 = time.perf_counter()
            for i in range(args.steps):
                model(batches[i % 2])
            dt = time.perf_counter() - t0
        print(json.dumps({"value": args.images * args.steps / dt, "unit": "images/s", "cores": args.threads, "kind": "port",
                          "sample": f"{args.steps} inference batches of {args.images} images {args.size}x{args.size}, torch "
                                    f"float32 eager on the host, {dt:.1f} s"}))
        return
    model.train()
    opt = solver.build_optimizer(cfg, model)
    step = solver.TrainStep(cfg, model, opt, world_size=1)
    batches = [syn.make_batch(args.images, 1234 + i, size=args.size) for i in range(2)]
    if args.weak:
        batches = [syn.add_scene_maps(b, 99 + i, ground_every=2) for i, b in enumerate(batches)]
    with d2.EventStorage(0):
        for i in range(args.warmup):
            step(batches[i % 2])
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(batches[i % 2])
        dt = time.perf_counter() - t0
        rep = step.report()
    print(json.dumps({"value": args.images * args.steps / dt, "unit": "images/s", "cores": args.threads, "kind": "port",
                      "sample": f"{args.steps} {'weakly supervised ' if args.weak else ''}train steps of {args.images} images "
                                f"{args.size}x{args.size}, torch float32 eager on the host, {dt:.1f} s",
                      "final_loss": rep["total_loss"]}))


if __name__ == "__main__":
    main()
