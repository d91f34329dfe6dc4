"""per-module isolation: feed each trunk module's GPU inputs to the bf16-emulating CPU backend and compare outputs."""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bt = importlib.import_module("bench_train")
from oracle import cpu_backend
dev = torch.device("cuda:0")
cfg, model, opt, syn, solver = bt.build(dev)
dla = importlib.import_module("3dod_amd.cubercnn.modeling.backbone.dla")
modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
batch = syn.make_batch(2, 21, with_gt=False)
rec = {}
def mk(name):
    def hook(m, inp, out):
        rec[name] = ([i.detach().float().cpu() if torch.is_tensor(i) else i for i in inp], out.detach().float().cpu())
    return hook
hs = []
for name, m in model.backbone.bottom_up.named_modules():
    if isinstance(m, (dla.BasicBlock, dla.Root, dla._Project, dla._ConvLevel)):
        hs.append(m.register_forward_hook(mk(name)))
model.train()
with torch.no_grad():
    images, x = model.preprocess_image(batch)
    feats = model.backbone(x)
for h in hs: h.remove()
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
cpu_backend.install(); cpu_backend.EMULATE_BF16 = True
cfg_cpu = syn.make_cfg(overrides=["MODEL.DEVICE", "cpu", "VIS_PERIOD", 0, "log", False])
ref = modeling.build_model(cfg_cpu); ref.load_state_dict(sd); ref.train()
mods = dict(ref.backbone.bottom_up.named_modules())
for name, (inp, out) in rec.items():
    with torch.no_grad():
        args = [i for i in inp if torch.is_tensor(i)]
        o = mods[name](*args)
    e = float((o - out).norm() / (out.norm() + 1e-12)); mx = float((o - out).abs().max() / (out.abs().max() + 1e-12))
    print(f"{name:40s} {type(mods[name]).__name__:12s} in={[tuple(a.shape) for a in args]} l2={e:.4f} max={mx:.4f}")
