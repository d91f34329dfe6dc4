"""host <-> device copies of one BoxNet batch (bench.py --workload boxnet inputs): which host lines issue them"""
import collections, importlib, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
dev = torch.device("cuda:0")
held = {}
modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
orig_build = modeling.build_model
def build(cfg):
    m = orig_build(cfg)
    oi = m.inference
    def spy(batch, **kw):
        held["batch"], held["kw"] = batch, kw
        return oi(batch, **kw)
    m.inference = spy
    held["oi"] = oi
    return m
modeling.build_model = build
class A: pass
args = A(); args.warmup, args.steps, args.no_cpu_baseline = 2, 1, True
bench.bench_boxnet(args, 0, 1, dev)
agg = collections.Counter()
class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        ts = [a for a in list(args) + list((kwargs or {}).values()) if torch.is_tensor(a)]
        devs = {t.device.type for t in ts}
        if torch.is_tensor(out):
            devs.add(out.device.type)
        name = str(func)
        if len(devs) > 1 or "_local_scalar_dense" in name or ("tolist" in name):
            where = "?"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if "3dod_amd" in fr.filename:
                    where = f"{fr.filename.split('repo/')[-1]}:{fr.lineno} {fr.name}"
                    break
            agg[(name.replace("aten.", ""), where)] += 1
        return out
with torch.no_grad(), Count():
    held["oi"](held["batch"], **held["kw"])
for (n, w), c in agg.most_common():
    print(f"{c:4d}  {n:28s} {w}")
