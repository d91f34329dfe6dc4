"""count the torch ops and launches of one weak train step (host-bound path) with torch.profiler"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bt = importlib.import_module("bench_train")
d2 = importlib.import_module("3dod_amd.d2lite")
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
rh = model.roi_heads
orig = rh.weak_losses_flat
def wrapped(*a, **k):
    with record_function("WEAK_FLAT_FWD"):
        return orig(*a, **k)
rh.weak_losses_flat = wrapped
with d2.EventStorage(1):
    for i in range(5):
        step(batches[i % 4])
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for i in range(4):
            step(batches[i % 4])
        torch.cuda.synchronize()
ev = prof.key_averages()
tot_cpu = sum(e.self_cpu_time_total for e in ev)
print("total self cpu ms / step", tot_cpu / 4e3)
rows = sorted(ev, key=lambda e: -e.self_cpu_time_total)[:28]
for e in rows:
    print(f"{e.key[:48]:48s} n/step={e.count / 4:7.1f} self_cpu={e.self_cpu_time_total / 4e3:7.3f} ms  cuda={e.self_device_time_total / 4e3:7.3f} ms")
print("total ops/step", sum(e.count for e in ev) / 4)
