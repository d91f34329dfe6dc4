import importlib, sys, os, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = int(sys.argv[1]); eager_first = int(sys.argv[2])
bt = importlib.import_module("bench_train")
dev = torch.device("cuda:0")
cfg, model, opt, syn, solver = bt.build(dev)
batch = syn.make_batch(N, 33, with_gt=False)
model.train()
pg = model.proposal_generator
if eager_first:
    images, x = model.preprocess_image(batch)
    feats = model.backbone(x)
    logits, deltas = pg.rpn_head([feats[f] for f in pg.in_features])
    loss = sum((f.float() ** 2).mean() for f in feats.values()) + sum(l.mean() for l in logits)
    if eager_first == 2:
        loss.backward(); opt.collect_grads()
    del feats, logits, deltas, loss
    torch.cuda.synchronize()
runner = model.enable_graphs(batch)
images, u8 = model._stack_images(batch)
f2, l2, d2_ = runner(u8)
(sum((f.float() ** 2).mean() for f in f2.values())).backward()
torch.cuda.synchronize()
print("OK", N, eager_first, float(opt.flat_g.abs().sum()))
