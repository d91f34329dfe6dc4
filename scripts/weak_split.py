"""where the weak train step spends its time: wall time of the cube branch, the ground-normal fit and the rest."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bt = importlib.import_module("bench_train")
d2 = importlib.import_module("3dod_amd.d2lite")
score = importlib.import_module("3dod_amd.cubercnn.modeling.roi_heads.roi_heads_score")
W = importlib.import_module("3dod_amd.cubercnn.modeling.roi_heads.weak_losses")
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
T = {}
def timed(name, fn):
    def wrap(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        return r
    return wrap
rh = model.roi_heads
rh._forward_cube = timed("cube_branch", rh._forward_cube)
rh._forward_box = timed("box_branch", rh._forward_box)
rh.label_and_sample_proposals = timed("label_and_sample", rh.label_and_sample_proposals)
model.proposal_generator.forward = timed("rpn", model.proposal_generator.forward)
W.ground_normals = timed("  ground_normals", W.ground_normals)
W.z_search_loss = timed("  z_search", W.z_search_loss)
W.pseudo_gt_z_box = timed("  z_box_median", W.pseudo_gt_z_box)
W.pose_alignment_loss = timed("  pose_align", W.pose_alignment_loss)
with d2.EventStorage(1):
    for i in range(5):
        step(batches[i % 4])
    T.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    N = 20
    for i in range(N):
        step(batches[i % 4])
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
print(f"step {tot / N * 1e3:.2f} ms")
for k, v in T.items():
    print(f"{k:22s} {v / N * 1e3:7.2f} ms")
