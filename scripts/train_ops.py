"""which torch ops (and shapes) remain in the default train step -- to find removable fills / copies"""
import importlib, os, sys, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bt = importlib.import_module("bench_train")
d2 = importlib.import_module("3dod_amd.d2lite")
dev = torch.device("cuda:0")
cfg, model, opt, syn, solver = bt.build(dev)
batches = [syn.make_batch(4, 1234 + i) for i in range(4)]
for b in batches:
    for d in b:
        d["image"], d["instances"] = d["image"].to(dev), d["instances"].to(dev)
step = solver.TrainStep(cfg, model, opt, world_size=1)
model.enable_graphs(batches[0]); opt.zero_grad()
from torch.profiler import profile, ProfilerActivity
with d2.EventStorage(0):
    for i in range(5):
        step(batches[i % 4])
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
        step(batches[0])
        torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::zeros", "aten::clone", "aten::contiguous", "aten::cat", "aten::to", "aten::_to_copy", "aten::mul", "aten::index")
cnt = collections.Counter()
for e in prof.events():
    if e.name in want and e.cpu_parent is not None:
        st = [s for s in (e.stack or []) if "3dod_amd" in s or "bench" in s]
        where = st[0].split("/")[-1] if st else "?"
        cnt[(e.name, str(e.input_shapes)[:60], where[:70])] += 1
for (name, shp, where), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:4d} {name:18s} {shp:62s} {where}")
