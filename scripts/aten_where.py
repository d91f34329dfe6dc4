"""which host lines launch the ATen kernels of one supervised train step (eager, current precision mode): a
TorchDispatchMode counts every aten op that touches a device tensor and attributes it to the innermost Python frame
inside this repo (backward ops of torch's own autograd nodes have no Python frame: 'autograd engine')"""
import collections, importlib, os, sys, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CR_GRAPHS", "0")
from torch.utils._python_dispatch import TorchDispatchMode
bt = importlib.import_module("bench_train")
d2 = importlib.import_module("3dod_amd.d2lite")
dev = torch.device("cuda:0")
cfg, model, opt, syn, solver = bt.build(dev)
batches = [syn.make_batch(4, 777 + i) for i in range(4)]
for b in batches:
    for d in b:
        for k in ("image", "instances"):
            d[k] = d[k].to(dev)
step = solver.TrainStep(cfg, model, opt, world_size=1)
opt.zero_grad()
SKIP = ("view", "reshape", "as_strided", "detach", "alias", "expand", "permute", "transpose", "t.default", "slice", "select",
        "unsqueeze", "squeeze", "_unsafe_view", "empty", "unbind", "split", "narrow", "unflatten", "size", "stride", "is_", "sym_",
        "_local_scalar_dense", "lift_fresh", "new_empty", "chunk", "flatten", "result_type", "set_", "record_stream")
agg = collections.defaultdict(lambda: [0, 0])
shapes = collections.Counter()
zshapes = collections.Counter()


class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        short = name.replace("aten.", "")
        if any(short.startswith(s) for s in SKIP):
            return out
        ts = [a for a in list(args) + list((kwargs or {}).values()) if torch.is_tensor(a)]
        if torch.is_tensor(out):
            ts.append(out)
        if not any(t.is_cuda for t in ts):
            return out
        where = "autograd engine"
        for fr in reversed(traceback.extract_stack(limit=30)):
            if ("3dod_amd" in fr.filename or "bench_train" in fr.filename) and "aten_where" not in fr.filename:
                where = f"{fr.filename.split('repo/')[-1]}:{fr.lineno} {fr.name}"
                break
        if where == 'autograd engine' and short.startswith('add.Tensor'):
            shapes[tuple(out.shape)] += 1
        if where == 'autograd engine' and short.startswith('zeros'):
            zshapes[tuple(out.shape)] += 1
        a = agg[(short, where)]
        a[0] += 1
        a[1] += max((t.numel() for t in ts), default=0)
        return out


NSTEP = 2
with d2.EventStorage(1):
    for i in range(3):
        step(batches[i % 4])
    torch.cuda.synchronize()
    with Count():
        for i in range(NSTEP):
            step(batches[i % 4])
    torch.cuda.synchronize()
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
print("aten ops on device tensors per step:", sum(v[0] for v in agg.values()) / NSTEP)
for (name, where), (n, el) in rows[:150]:
    print(f"n={n / NSTEP:5.1f} maxnumel/op={el / n:11.0f}  {name:28s} {where[:130]}")

print('autograd-engine fan-in adds by shape (per step):')
for sh, n in sorted(shapes.items(), key=lambda kv: -kv[1] * torch.Size(kv[0]).numel()):
    print(f'  {n / NSTEP:4.1f} x {sh}')

print('autograd-engine zero fills by shape (per step): materialised gradients of unused outputs')
for sh, n in sorted(zshapes.items(), key=lambda kv: -kv[1] * torch.Size(kv[0]).numel()):
    print(f'  {n / NSTEP:4.1f} x {sh}')
