"""phase-by-phase check of the whole-step graph at the bench batch size; progress goes to a file."""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LOG = open(os.path.join("gpurun_out", "wholestep_progress.log"), "w")
def say(*a):
    print(*a, file=LOG, flush=True); os.fsync(LOG.fileno())
bt = importlib.import_module("bench_train")
dev = torch.device("cuda:0")
B = int(os.environ.get("CR_B", "4"))
bt.IMS_PER_GPU = B
cfg, model, opt, syn, solver = bt.build(dev)
d2 = importlib.import_module("3dod_amd.d2lite")
dt = importlib.import_module("3dod_amd.cubercnn.modeling.dense_train")
batches = [syn.make_batch(B, 1234 + i) for i in range(4)]
for b in batches:
    for d in b:
        d["image"] = d["image"].to(dev); d["instances"] = d["instances"].to(dev)
say("built")
with d2.EventStorage(0):
    # phase 1: the static path eagerly, G padded to 32, different batches
    for i in range(8):
        data = batches[i % 4]
        images, u8 = model._stack_images(data)
        sizes = [tuple(s) for s in images.image_sizes]
        gt = dt.GTBatch([d["instances"] for d in data], dev, G=32)
        meta = dt.camera_meta(model.roi_heads, [torch.as_tensor(d["K"]) for d in data], [1.0] * B, sizes, dev)
        loss = sum(model.forward_static(u8, sizes, gt, meta).values())
        opt.zero_grad(); loss.backward(); opt.collect_grads(); opt.step()
        torch.cuda.synchronize()
        say("eager static step", i, float(loss))
    step = solver.GraphedTrainStep(cfg, model, opt, batches[0])
    torch.cuda.synchronize()
    say("captured")
    for i in range(12):
        step.load(batches[i % 4]); torch.cuda.synchronize(); say("loaded", i)
        step.graph_a.replay(); torch.cuda.synchronize(); say("A", i)
        step.graph_b.replay(); torch.cuda.synchronize(); say("B", i, step.report()["total_loss"])
say("done")
