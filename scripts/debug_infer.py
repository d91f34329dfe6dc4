import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bt = importlib.import_module("bench_train")
dev = torch.device("cuda:0")
cfg, model, opt, syn, solver = bt.build(dev)
d2 = importlib.import_module("3dod_amd.d2lite")
step = solver.TrainStep(cfg, model, opt)
with d2.EventStorage(0):
    for _ in range(3):
        step(syn.make_batch(2, 7))
print(step.report())
model.eval()
batch = syn.make_batch(2, 9, with_gt=False)
with torch.no_grad():
    images, x = model.preprocess_image(batch)
    print("x finite", torch.isfinite(x.float()).all().item(), x.shape)
    feats = model.backbone(x)
    for k, v in feats.items():
        print(k, tuple(v.shape), "finite", torch.isfinite(v.float()).all().item(), float(v.float().abs().max()))
    props, _ = model.proposal_generator(images, feats, None)
    for p in props:
        print("proposals", len(p), p.objectness_logits[:3])
    rh = model.roi_heads
    f = [feats[k] for k in rh.box_in_features]
    bf = rh.box_pooler(f, [p.proposal_boxes for p in props])
    print("pooled finite", torch.isfinite(bf.float()).all().item())
    h = rh.box_head(bf)
    print("head finite", torch.isfinite(h.float()).all().item())
    scores, deltas = rh.box_predictor(h)
    print("scores finite", torch.isfinite(scores).all().item(), "deltas finite", torch.isfinite(deltas).all().item())
    probs = torch.softmax(scores, -1)
    print("max fg prob", float(probs[:, :-1].max()), "thresh", rh.box_predictor.test_score_thresh)
    boxes = rh.box_predictor.predict_boxes((scores, deltas), props)
    print("boxes finite", torch.isfinite(boxes[0]).all().item())
    rh.box_predictor.test_score_thresh = 0.0
    inst, _ = rh.box_predictor.inference((scores, deltas), props)
    print([len(i) for i in inst])
