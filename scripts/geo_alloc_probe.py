"""does the fast project+score kernel's duration depend on where its output planes live?  fast-only process, fresh
allocations per call / one reused set / one reused set carved out of ONE buffer at staggered offsets"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
geo = importlib.import_module("3dod_amd.geometry")
dev = "cuda:0"
inp = bench.geometry_inputs(1024, 1000, 1234, dev)
a = (inp["cubes"], inp["K"], inp["im_wh"], inp["ref"], inp["mu"], inp["sg"], inp["rect"])
def t(fn, n=200):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"fresh allocations per call: {t(lambda: geo.cubes_project_score(*a, fast=True)):.1f} us")
o = geo.cubes_project_score(*a, fast=True)
print(f"one reused set:             {t(lambda: geo.cubes_project_score(*a, fast=True, out=o)):.1f} us")
print("bases mod 1 MiB:", {k: (v.data_ptr() >> 12) & 255 for k, v in o.items() if v is not None})
for stagger in (0, 4096 + 256, 65536 + 4096, 1 << 20):
    N, P = 1024, 1000
    sizes = {"corners": N * P * 16, "boxes": N * P * 4, "iou": N * P, "dim": N * P, "corner": N * P, "combined": N * P}
    big = torch.empty(sum(sizes.values()) + 32 * (stagger // 4 + 1024), device=dev)
    off, o2 = 0, {}
    shp = {"corners": (N, P, 8, 2), "boxes": (N, P, 4), "iou": (N, P), "dim": (N, P), "corner": (N, P), "combined": (N, P)}
    for i, (k, n) in enumerate(sizes.items()):
        o2[k] = big[off:off + n].view(shp[k]); off += n + stagger // 4 * (i + 1)
        off = (off + 63) // 64 * 64
    o2["argmax"], o2["best"] = o["argmax"], o["best"]
    print(f"one buffer, planes staggered by {stagger:8d} B: {t(lambda: geo.cubes_project_score(*a, fast=True, out=o2)):.1f} us")
