"""kernel launches per section of the weak cube branch (forward), and of the whole backward"""
import importlib, os, sys, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bt = importlib.import_module("bench_train")
d2 = importlib.import_module("3dod_amd.d2lite")
W = importlib.import_module("3dod_amd.cubercnn.modeling.roi_heads.weak_losses")
util = importlib.import_module("3dod_amd.cubercnn.util.math_util")
dev = torch.device("cuda:0")
cfg, model, opt, syn, solver = bt.build(dev, config="Omni_combined.yaml", lr=0.0012)
batches = [syn.add_scene_maps(syn.make_batch(2, 777 + i), 99 + i, ground_every=2) for i in range(4)]
for b in batches:
    for d in b:
        for k in ("image", "instances", "depth_map"):
            d[k] = d[k].to(dev)
        if d["ground_map"] is not None:
            d["ground_map"] = d["ground_map"].to(dev)
step = solver.TrainStep(cfg, model, opt, world_size=1)
model.enable_graphs(batches[0]); opt.zero_grad()
from torch.profiler import profile, ProfilerActivity, record_function
def mark(mod, name):
    f = getattr(mod, name)
    def w(*a, **k):
        with record_function("SEC_" + name):
            return f(*a, **k)
    setattr(mod, name, w)
for n in ("project_cubes_to_corners", "generalized_box_iou_loss", "pose_alignment_loss", "ground_normals", "z_search_loss",
          "pseudo_gt_z_box", "dim_hinge_loss", "corners_to_boxes"):
    mark(W, n)
for n in ("R_from_allocentric", "get_cuboid_verts_faces"):
    mark(util, n)
rh = model.roi_heads
for n in ("weak_losses_flat", "safely_reduce_losses"):
    f = getattr(rh, n)
    def mk(f, n):
        def w(*a, **k):
            with record_function("SEC_" + n):
                return f(*a, **k)
        return w
    setattr(rh, n, mk(f, n))
with d2.EventStorage(1):
    for i in range(4):
        step(batches[i % 4])
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        step(batches[0])
        torch.cuda.synchronize()
evs = prof.events()
secs = [e for e in evs if e.name.startswith("SEC_")]
launch = [e for e in evs if e.name in ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipMemcpyWithStream", "hipMemcpyAsync")]
cnt = collections.Counter(); dur = collections.Counter()
for s in secs:
    lo, hi = s.time_range.start, s.time_range.end
    n = sum(1 for l in launch if lo <= l.time_range.start <= hi)
    cnt[s.name] += n; dur[s.name] += (hi - lo)
for k in sorted(cnt, key=lambda k: -cnt[k]):
    print(f"{k:32s} launches={cnt[k]:5d}  host_ms={dur[k]/1e3:7.2f}")
print("all launches in the step:", len(launch))
