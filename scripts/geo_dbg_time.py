import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
geo = importlib.import_module("3dod_amd.geometry")
dev = "cuda:0"
inp = bench.geometry_inputs(1024, 1000, 1234, dev)
a = (inp["cubes"], inp["K"], inp["im_wh"], inp["ref"], inp["mu"], inp["sg"], inp["rect"])
ALL = ("corners", "boxes", "iou", "dim", "corner", "combined")
for want in (ALL, ()):
    for _ in range(20):
        geo.cubes_project_score(*a, want=want, fast=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        geo.cubes_project_score(*a, want=want, fast=True)
    e1.record(); torch.cuda.synchronize()
    print(f"dbg={os.environ.get('CR_GEO_DBG')} want={'all' if want else 'none'}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us")
