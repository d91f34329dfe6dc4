import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bt = importlib.import_module("bench_train")
dev = torch.device("cuda:0")
cfg, model, opt, syn, solver = bt.build(dev)
batches = [syn.make_batch(4, 1234 + i, with_gt=False) for i in range(2)]
r = model.enable_graphs(batches[0]); opt.zero_grad()
g = torch.Generator(device=dev).manual_seed(0)
def fwd(b):
    il, u8 = model._stack_images(b)
    r.static_img.copy_(u8); r.fwd_graph.replay(); torch.cuda.synchronize()
    return [o.clone() for o in r.static_outs]
def bwd():
    opt.flat_g.zero_(); r.bwd_graph.replay(); torch.cuda.synchronize()
    return opt.flat_g.clone()
for sg in r.static_grads:
    sg.copy_(torch.randn(sg.shape, device=dev, generator=g).to(sg.dtype) * 1e-3)
o1 = fwd(batches[0]); g1 = bwd(); g1b = bwd()
print("bwd idempotent (same fwd):", float((g1 - g1b).abs().max()), float(g1.abs().max()))
o2 = fwd(batches[0]); g2 = bwd()
print("fwd replay twice equal:", all(torch.equal(a, b) for a, b in zip(o1, o2)), "bwd after 2nd fwd:", float((g1 - g2).abs().max()))
o3 = fwd(batches[1]); g3 = bwd(); print("other batch: finite", bool(torch.isfinite(g3).all()), float(g3.abs().max()))
# now perturb weights like an optimizer step and bump epoch
opt.flat_p.mul_(1.001); importlib.import_module("3dod_amd.hipops").bump_weight_epoch()
o4 = fwd(batches[0]); g4 = bwd(); print("after weight change: finite", bool(torch.isfinite(g4).all()), float(g4.abs().max()), float((g4-g1).abs().max()))
# which parameters blow up for the other batch?
names = {id(p): n for n, p in model.named_parameters()}
o3 = fwd(batches[1]); g3 = bwd()
rows = []
for p in opt.params:
    v = p._cr_grad
    rows.append((float(v.abs().max()), names.get(id(p), "?"), tuple(p.shape)))
rows.sort(reverse=True)
for r_ in rows[:12]:
    print("%.3e %s %s" % r_)
print("n huge:", sum(1 for r_ in rows if r_[0] > 1e6), "of", len(rows))
