"""wall time per region of the train step (synchronised at region boundaries)."""
import importlib, sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bt = importlib.import_module("bench_train")
dev = torch.device("cuda:0")
cfg, model, opt, syn, solver = bt.build(dev)
d2 = importlib.import_module("3dod_amd.d2lite")
batches = [syn.make_batch(4, 1234 + i) for i in range(2)]
for b in batches:
    for d in b:
        d["image"] = d["image"].to(dev); d["instances"] = d["instances"].to(dev)
if os.environ.get("CR_NO_GRAPHS", "0") != "1":
    model.enable_graphs(batches[0]); opt.zero_grad()
T = {}
def tic():
    torch.cuda.synchronize(); return time.perf_counter()
def lap(name, t0):
    torch.cuda.synchronize(); T[name] = T.get(name, 0) + time.perf_counter() - t0
rh = model.roi_heads; pg = model.proposal_generator
N = 10
with d2.EventStorage(0):
    for it in range(N + 3):
        if it == 3: T.clear()
        data = batches[it % 2]
        t = tic()
        g = model._graphed
        images, batch = model._stack_images(data)
        if g is not None:
            features, logits, deltas = g(batch); ho = (logits, deltas)
        else:
            x = importlib.import_module("3dod_amd.hipops").preprocess(batch, model.pixel_mean_list, model.pixel_std_list)
            features = model.backbone(x); ho = None
        lap("dense_fwd", t); t = tic()
        gt = [b["instances"] for b in data]
        Ks = [torch.FloatTensor(b["K"]) for b in data]; ratios = [1.0] * len(data)
        proposals, pl = pg(images, features, gt, head_outputs=ho)
        lap("rpn(label+loss+proposals)", t); t = tic()
        props = rh.label_and_sample_proposals(proposals, gt)
        lap("roi_label_sample", t); t = tic()
        lb = rh._forward_box(features, props)
        lap("box_head+loss", t); t = tic()
        inst, lc = rh._forward_cube(features, props, Ks, [(512, 512)] * len(data), ratios)
        lap("cube_head+loss", t); t = tic()
        loss = sum(pl.values()) + sum(lb.values()) + sum(lc.values())
        opt.zero_grad(); loss.backward(); opt.collect_grads()
        lap("backward", t); t = tic()
        opt.step()
        lap("sgd", t)
tot = sum(T.values())
for k, v in T.items():
    print(f"{k:28s} {v / N * 1e3:8.2f} ms")
print(f"{'sum':28s} {tot / N * 1e3:8.2f} ms")
