"""exact vs fast project+score kernel on the bench's geometry inputs: time per launch, fallback / candidate counters"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
geo = importlib.import_module("3dod_amd.geometry")
dev = "cuda:0"
inp = bench.geometry_inputs(1024, 1000, 1234, dev)
a = (inp["cubes"], inp["K"], inp["im_wh"], inp["ref"], inp["mu"], inp["sg"], inp["rect"])
ALL = ("corners", "boxes", "iou", "dim", "corner", "combined")
st = torch.zeros(2, dtype=torch.int64, device=dev)
geo.cubes_project_score(*a, fast=True, stats=st)
print("stats: exact objects, candidates", st.tolist())
for want in (ALL, ()):
    for fast in (False, True):
        for _ in range(20):
            geo.cubes_project_score(*a, want=want, fast=fast)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            geo.cubes_project_score(*a, want=want, fast=fast)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        b = 156 if want else 60
        print(f"want={'all' if want else 'none'} fast={fast}: {us:.1f} us  {b * 1.024e6 / us / 1e6:.2f} TB/s  frac {b * 1.024e6 / us / 1e6 / 8:.3f}")
