"""fast / exact project+score kernel per requested output set (which stores cost what)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
geo = importlib.import_module("3dod_amd.geometry")
dev = "cuda:0"
inp = bench.geometry_inputs(1024, 1000, 1234, dev)
a = (inp["cubes"], inp["K"], inp["im_wh"], inp["ref"], inp["mu"], inp["sg"], inp["rect"])
B = {"corners": 64, "boxes": 16, "iou": 4, "dim": 4, "corner": 4, "combined": 4}
for want in ((), ("corners",), ("boxes",), ("iou", "dim", "corner", "combined"), ("boxes", "iou", "dim", "corner", "combined"),
             ("corners", "boxes"), tuple(B)):
    for fast in (True, False):
        for _ in range(20):
            geo.cubes_project_score(*a, want=want, fast=fast)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            geo.cubes_project_score(*a, want=want, fast=fast)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        b = 60 + sum(B[k] for k in want)
        print(f"fast={fast!s:5} {b:4d} B/cube {us:6.1f} us {b * 1.024e6 / us / 1e6:5.2f} TB/s  want={','.join(want) or '-'}")
