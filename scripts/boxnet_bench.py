"""BoxNet (1000-cube proposal-and-scoring) end to end on GT boxes: images/s for a batch of 8 images x 4..16 objects"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
syn = importlib.import_module("3dod_amd.synthetic")
modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
dev = torch.device("cuda:0")
cfg = syn.make_cfg(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "BoxNet.yaml"),
                   ["MODEL.DEVICE", "cuda:0", "VIS_PERIOD", 0, "log", False])
torch.manual_seed(0)
model = modeling.build_model(cfg).eval()
B = 8
batch = syn.make_batch(B, 5)
g = torch.Generator().manual_seed(2)
for b in batch:
    b["image"] = b["image"].to(dev)
    b["instances"] = b["instances"].to(dev)
    b["depth_map"] = (torch.rand(512, 512, generator=g) * 3 + 1).to(dev)
    b["ground_map"] = (torch.arange(512)[:, None] > 300).expand(512, 512).to(torch.uint8).to(dev)
    n = len(b["instances"])
    m = torch.zeros(n, 512, 512, dtype=torch.bool)
    for j, bb in enumerate(b["instances"].gt_boxes.tensor.round().long().clamp(0, 511).cpu()):
        m[j, bb[1]:bb[3] + 1, bb[0]:bb[2] + 1] = True
    b["masks"] = m.to(dev)
gen = torch.Generator(device=dev).manual_seed(3)
nobj = sum(len(b["instances"]) for b in batch)
for _ in range(70):            # hipops keeps the last 128 argument tensors alive: 64 calls until the allocator is steady
    model.inference(batch, experiment_type={"use_pred_boxes": False}, generator=gen)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 50
for _ in range(N):
    model.inference(batch, experiment_type={"use_pred_boxes": False}, generator=gen)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f"BoxNet on GT boxes: {B} images, {nobj} objects x 1000 cubes: {dt * 1e3:.2f} ms per batch = {B / dt:.0f} images/s, {nobj * 1000 / dt / 1e6:.1f} M cubes/s")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
model.inference(batch, experiment_type={"use_pred_boxes": False}, generator=gen)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
