"""cr_box_median on the weak step's shapes: 2 x 512 x 512 smooth depth maps, 256 windows"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
geo = importlib.import_module("3dod_amd.geometry")
syn = importlib.import_module("3dod_amd.synthetic")
dev = torch.device("cuda:0")
b = syn.add_scene_maps(syn.make_batch(2, 1), 3)
depth = torch.stack([d["depth_map"] for d in b]).to(dev)
g = torch.Generator().manual_seed(0)
c = torch.rand(256, 2, generator=g) * 512
wh = torch.rand(256, 2, generator=g) * 300 + 20
boxes = torch.cat((c - wh / 2, c + wh / 2), 1).clamp(0, 512).long().to(torch.int32).to(dev)
img = torch.randint(0, 2, (256,), generator=g).to(torch.int32).to(dev)
for _ in range(5):
    out = geo.box_median(depth, boxes, img)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    out = geo.box_median(depth, boxes, img)
e1.record(); torch.cuda.synchronize()
print("box_median us/call", e0.elapsed_time(e1) / 20 * 1e3)
