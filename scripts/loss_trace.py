import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bt = importlib.import_module("bench_train")
dev = torch.device("cuda:0")
cfg, model, opt, syn, solver = bt.build(dev, lr=float(os.environ['CR_LR']) if 'CR_LR' in os.environ else None)
d2 = importlib.import_module("3dod_amd.d2lite")
batches = [syn.make_batch(4, 1234 + i) for i in range(4)]
for b in batches:
    for d in b:
        d["image"] = d["image"].to(dev); d["instances"] = d["instances"].to(dev)
if os.environ.get("CR_NO_GRAPHS", "0") != "1":
    model.enable_graphs(batches[0]); opt.zero_grad()
step = solver.TrainStep(cfg, model, opt)
with d2.EventStorage(0):
    for i in range(int(os.environ.get('CR_STEPS', '12'))):
        step(batches[i % 4])
        r = step.report()
        gn = float(opt.flat_g.norm())
        if i % 5 == 4 or r['iterations_explode'] > 0:
            print(i, f"total={r['total_loss']:.4g} skipped={r['iterations_explode']:.0f} gradnorm={gn:.4g}",
                  {k: round(v, 3) for k, v in r.items() if '/' in k})
