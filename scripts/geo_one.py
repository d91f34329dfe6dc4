"""one variant of the project+score kernel, 200 launches (for rocprofv3 --kernel-trace --stats): argv = fast|exact all|none"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
geo = importlib.import_module("3dod_amd.geometry")
dev = "cuda:0"
fast = sys.argv[1] == "fast"
want = ("corners", "boxes", "iou", "dim", "corner", "combined") if sys.argv[2] == "all" else ()
inp = bench.geometry_inputs(1024, 1000, 1234, dev)
a = (inp["cubes"], inp["K"], inp["im_wh"], inp["ref"], inp["mu"], inp["sg"], inp["rect"])
for _ in range(220):
    geo.cubes_project_score(*a, want=want, fast=fast)
torch.cuda.synchronize()
