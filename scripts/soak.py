"""soak: N train steps on cycling synthetic batches; reports loss trajectory, skipped steps and device-memory growth"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bt = importlib.import_module("bench_train")
d2 = importlib.import_module("3dod_amd.d2lite")
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg, model, opt, syn, solver = bt.build(dev)
batches = [syn.make_batch(4, 1234 + i) for i in range(16)]
for b in batches:
    for d in b:
        d["image"], d["instances"] = d["image"].to(dev), d["instances"].to(dev)
step = solver.TrainStep(cfg, model, opt, world_size=1)
model.enable_graphs(batches[0]); opt.zero_grad()
sched = solver.WarmupMultiStepLR(opt, [], 0.1, 0.001, 100, "linear", None)
with d2.EventStorage(0):
    for i in range(10):
        step(batches[i % 16]); sched.step()
    torch.cuda.synchronize()
    m0 = torch.cuda.memory_allocated(dev)
    t0 = time.perf_counter()
    for i in range(N):
        step(batches[i % 16]); sched.step()
        if (i + 1) % 100 == 0:
            r = step.report()
            print(f"iter {i + 1}: total_loss {r['total_loss']:.3f} skipped {r['iterations_explode']:.0f} mem {torch.cuda.memory_allocated(dev) / 2**20:.0f} MiB", flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"{N} steps, {dt / N * 1e3:.2f} ms/step, memory growth {(torch.cuda.memory_allocated(dev) - m0) / 2**20:.1f} MiB")
