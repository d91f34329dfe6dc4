"""host (enqueue) time of the inference step, section by section (GPU drained before every step)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
bt = importlib.import_module("bench_train")
rcnn = importlib.import_module("3dod_amd.cubercnn.modeling.meta_arch.rcnn3d")
rh_mod = importlib.import_module("3dod_amd.cubercnn.modeling.roi_heads.roi_heads")
fr = importlib.import_module("3dod_amd.cubercnn.modeling.roi_heads.fast_rcnn")
rpn = importlib.import_module("3dod_amd.cubercnn.modeling.proposal_generator.rpn")
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
dev = torch.device("cuda", 0)
cfg, model, opt, syn, solver = bt.build(dev)
model.eval()
wrap(rcnn.RCNN3D, "_postprocess"); wrap(rcnn.RCNN3D, "_stack_images"); wrap(rcnn.RCNN3D, "_run_roi_heads")
wrap(type(model.proposal_generator), "forward", "proposal_generator")
wrap(rh_mod.ROIHeads3D, "_forward_box"); wrap(rh_mod.ROIHeads3D, "_forward_cube")
wrap(fr, "fast_rcnn_inference")
for n in ("find_top_rpn_proposals",):
    if hasattr(rpn, n):
        wrap(rpn, n)
batches = [syn.make_batch(8, 4321 + i, with_gt=False) for i in range(2)]
for b in batches:
    for d in b:
        d["image"] = d["image"].to(dev)
with torch.no_grad():
    model.enable_graphs_eval(batches[0])
    for i in range(4):
        model(batches[i % 2])
    acc.clear()
    tot = 0.0
    N = 10
    for i in range(N):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model(batches[i % 2])
        tot += time.perf_counter() - t0
print(f"host time per inference batch (incl. its two host waits) {tot / N * 1e3:.3f} ms")
for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:32s} {t / N * 1e3:8.3f} ms/step  ({n / N:.1f} calls)")
