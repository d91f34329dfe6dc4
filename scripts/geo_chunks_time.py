"""fast project+score kernel: time against the number of 256-cube chunks per object (slope = per chunk, intercept = fixed)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
geo = importlib.import_module("3dod_amd.geometry")
dev = "cuda:0"
ALL = ("corners", "boxes", "iou", "dim", "corner", "combined")
for P in (256, 512, 768, 1024):
    inp = bench.geometry_inputs(1024, P, 1234, dev)
    a = (inp["cubes"], inp["K"], inp["im_wh"], inp["ref"], inp["mu"], inp["sg"], inp["rect"])
    for want in ((), ALL):
        for fast in (True, False):
            for _ in range(20):
                geo.cubes_project_score(*a, want=want, fast=fast)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                geo.cubes_project_score(*a, want=want, fast=fast)
            e1.record(); torch.cuda.synchronize()
            print(f"P={P:5d} want={'all ' if want else 'none'} fast={fast!s:5}: {e0.elapsed_time(e1) / 200 * 1e3:6.1f} us")
