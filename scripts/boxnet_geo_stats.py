"""how the project+score launch of the BoxNet pipeline (bench.py --workload boxnet inputs) splits into fast / exact objects"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
geo = importlib.import_module("3dod_amd.geometry")
dev = torch.device("cuda:0")
calls = []
orig = geo.cubes_project_score
def spy(*a, **kw):
    if kw.get("want", None) == ():
        st = torch.zeros(2, dtype=torch.int64, device=dev)
        kw["stats"] = st
        kw["fast"] = True
        r = orig(*a, **kw)
        calls.append((st, a[0].shape, a))
        return r
    return orig(*a, **kw)
geo.cubes_project_score = spy
class A: pass
args = A(); args.warmup, args.steps, args.no_cpu_baseline = 1, 2, True
res = bench.bench_boxnet(args, 0, 1, dev)
torch.cuda.synchronize()
for st, shp, a in calls[:3]:
    print("objects x cubes", tuple(shp[:2]), "exact objects", int(st[0]), "candidates", int(st[1]))
st, shp, a = calls[-1]
cubes, K, im, ref, mu, sg, rect = a[:7]
ex = orig(cubes, K, im, ref, mu, sg, rect, fast=False)
fa = orig(cubes, K, im, ref, mu, sg, rect, fast=True)
pois = ~torch.isfinite(ex["combined"]).all(1)
print("objects with a non-finite exact score:", int(pois.sum()), " best < 1e-6:", int((ex["best"] < 1e-6).sum()),
      " boxes thinner than 1 px with overlap:", int((((ex["boxes"][..., 2] - ex["boxes"][..., 0]).clamp_max(1e9) < 1) & (ex["iou"] > 0)).any(1).sum()),
      " or flatter:", int((((ex["boxes"][..., 3] - ex["boxes"][..., 1]) < 1) & (ex["iou"] > 0)).any(1).sum()))
print("argmax equal:", bool(torch.equal(ex["argmax"], fa["argmax"])))
def t(fn, n=100):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
o = orig(cubes, K, im, ref, mu, sg, rect, want=(), fast=True)
print(f"argmax-only launch on these inputs: fast {t(lambda: orig(cubes, K, im, ref, mu, sg, rect, want=(), fast=True, out=o)):.1f} us, "
      f"exact {t(lambda: orig(cubes, K, im, ref, mu, sg, rect, want=(), fast=False, out=o)):.1f} us")
