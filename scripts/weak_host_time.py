"""host (enqueue) time of the weak train step, section by section: the GPU is drained before every step, so the wall clock of a
section is the time the host needs to enqueue it (nothing inside the step waits for the device).
    python scripts/weak_host_time.py [weak|train]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
which = sys.argv[1] if len(sys.argv) > 1 else "weak"
bt = importlib.import_module("bench_train")
d2 = importlib.import_module("3dod_amd.d2lite")
dense = importlib.import_module("3dod_amd.cubercnn.modeling.dense_train")
ops = importlib.import_module("3dod_amd.hipops")
W = importlib.import_module("3dod_amd.cubercnn.modeling.roi_heads.weak_losses")
solver_build = importlib.import_module("3dod_amd.cubercnn.solver.build")
acc = {}
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        e = acc.setdefault(label or name, [0.0, 0])
        e[0] += time.perf_counter() - t0; e[1] += 1
        return r
    setattr(obj, name, g)
for n in ("rpn_label_and_sample", "rpn_losses", "rpn_proposals_padded", "roi_label_and_sample", "pool_roi_features", "box_head_losses",
          "weak_cube_losses_fused", "cube_head_losses", "rpn_tensors", "forward_train_weak", "forward_train"):
    wrap(dense, n)
wrap(W, "ground_normals_batched")
wrap(ops, "weak_cube_loss")
dev = torch.device("cuda", 0)
B = 2 if which == "weak" else 4
cfg, model, opt, syn, solver = bt.build(dev, world=1, config="Omni_combined.yaml" if which == "weak" else "Base_Omni3D.yaml", lr=0.001)
mk = (lambda i: syn.add_scene_maps(syn.make_batch(B, 777 + i), 99 + i, ground_every=2)) if which == "weak" else (lambda i: syn.make_batch(B, 777 + i))
batches = [mk(i) for i in range(4)]
for b in batches:
    for d in b:
        for k in ("image", "instances", "depth_map", "ground_map"):
            if d.get(k) is not None:
                d[k] = d[k].to(dev)
step = solver.TrainStep(cfg, model, opt, world_size=1)
model.enable_graphs(batches[0])
opt.zero_grad()
wrap(type(model), "forward", "model.forward")
wrap(torch.Tensor, "backward", "backward")
wrap(type(opt), "step", "optimizer.step")
tot = 0.0
with d2.EventStorage(1):
    for i in range(8):
        step(batches[i % 4])
    acc.clear()
    N = 20
    for i in range(N):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(batches[i % 4])
        tot += time.perf_counter() - t0
torch.cuda.synchronize()
print(f"{which}: host time per step {tot / N * 1e3:.3f} ms")
for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:32s} {t / N * 1e3:8.3f} ms/step  ({n / N:.1f} calls)")
# free-running: per-step host time without draining the GPU (a step that takes as long as the GPU step means the host waited inside it)
acc.clear()
ts = []
with d2.EventStorage(1):
    torch.cuda.synchronize()
    for i in range(24):
        t0 = time.perf_counter()
        step(batches[i % 4])
        ts.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    tail = (time.perf_counter() - t0) * 1e3
print("free-running host ms per step:", " ".join(f"{t:.1f}" for t in ts), "| drain %.1f ms" % tail)
for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:8]:
    print(f"  {k:32s} {t / 24 * 1e3:8.3f} ms/step")
