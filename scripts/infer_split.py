import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bt = importlib.import_module("bench_train"); d2 = importlib.import_module("3dod_amd.d2lite")
dev = torch.device("cuda:0")
cfg, model, opt, syn, solver = bt.build(dev)
model.eval()
b = syn.make_batch(8, 1, with_gt=False)
for d in b: d["image"] = d["image"].to(dev)
def T(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad(), d2.EventStorage(0):
    images, x = model.preprocess_image(b)
    feats = model.backbone(x)
    props, _ = model.proposal_generator(images, feats, None)
    Ks = [torch.FloatTensor(i['K']) for i in b]; r = [1.0] * 8
    print("preprocess", T(lambda: model.preprocess_image(b)))
    print("backbone", T(lambda: model.backbone(x)))
    print("rpn", T(lambda: model.proposal_generator(images, feats, None)))
    print("roi_heads", T(lambda: model.roi_heads(images, feats, props, Ks, r, None)))
    print("total", T(lambda: model(b)))
    rh = model.roi_heads
    print("  _forward_box", T(lambda: rh._forward_box(feats, props)))
    inst = rh._forward_box(feats, props)
    print("  _forward_cube", T(lambda: rh._forward_cube(feats, inst, Ks, [im.shape[-2:] for im in [d["image"] for d in b]], r)))
