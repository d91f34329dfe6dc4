"""end-to-end sanity of the training step: memorise two synthetic batches (the total loss must fall well below its start)"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bt = importlib.import_module("bench_train")
d2 = importlib.import_module("3dod_amd.d2lite")
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
cfg, model, opt, syn, solver = bt.build(dev, lr=float(sys.argv[2]) if len(sys.argv) > 2 else 0.0025)
batches = [syn.make_batch(4, 99 + i, min_obj=3, max_obj=6) for i in range(2)]
for b in batches:
    for d in b:
        d["image"], d["instances"] = d["image"].to(dev), d["instances"].to(dev)
step = solver.TrainStep(cfg, model, opt, world_size=1)
model.enable_graphs(batches[0]); opt.zero_grad()
sched = solver.WarmupMultiStepLR(opt, [], 0.1, 0.01, 50, "linear", None)
with d2.EventStorage(0):
    for i in range(N):
        step(batches[i % 2]); sched.step()
        if i in (0, 9) or (i + 1) % 100 == 0:
            r = step.report()
            keys = ["total_loss", "rpn/cls", "rpn/loc", "BoxHead/loss_cls", "BoxHead/loss_box_reg", "Cube/loss_z", "Cube/loss_dims", "Cube/loss_pose"]
            print(f"iter {i + 1}: " + "  ".join(f"{k.split('/')[-1]} {r[k]:.3f}" for k in keys if k in r) + f"  skipped {r['iterations_explode']:.0f}", flush=True)
