"""Per-region kernel accounting of one eager static-shape train step.

run:    rocprofv3 --kernel-trace --output-format csv -d /tmp/rt -o rt -- python3 scripts/region_trace.py run
parse:  python3 scripts/region_trace.py parse /tmp/rt/rt_kernel_trace.csv > gpurun_out/region_trace.txt
Regions are delimited by marker fills of distinctive sizes (calibrated at the start of the trace)."""
import importlib, os, sys, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REGIONS = ["preprocess+trunk+fpn fwd", "rpn head fwd", "anchors+cat", "rpn label/sample", "rpn losses", "rpn proposals (topk+nms)",
           "roi label/sample", "box head fwd+loss", "cube head fwd+loss", "loss stack", "bwd: heads", "bwd: rpn+fpn+trunk",
           "grad collect+sgd", "end"]
BASE, STRIDE = 3_000_000, 262_144


def run():
    import torch
    bt = importlib.import_module("bench_train")
    dev = torch.device("cuda:0")
    cfg, model, opt, syn, solver = bt.build(dev)
    d2 = importlib.import_module("3dod_amd.d2lite")
    dt = importlib.import_module("3dod_amd.cubercnn.modeling.dense_train")
    ops = importlib.import_module("3dod_amd.hipops")
    B = bt.IMS_PER_GPU
    data = syn.make_batch(B, 1234)
    for d in data:
        d["image"] = d["image"].to(dev); d["instances"] = d["instances"].to(dev)
    on = [False]

    def mark(k):
        if on[0]:
            torch.zeros(BASE + STRIDE * k, device=dev)

    def wrap(mod, name, before, after=None):
        f = getattr(mod, name)
        def g(*a, **kw):
            mark(before)
            r = f(*a, **kw)
            if after is not None:
                mark(after)
            return r
        setattr(mod, name, g)
    wrap(dt, "rpn_label_and_sample", 3, 4)
    wrap(dt, "rpn_losses", 4, 5)
    wrap(dt, "rpn_proposals_padded", 5, 6)
    wrap(dt, "roi_label_and_sample", 6, 7)
    wrap(dt, "box_head_losses", 7, 8)
    wrap(dt, "cube_head_losses", 8, 9)
    head_fwd = model.proposal_generator.rpn_head.forward
    def head(*a, **kw):
        mark(1); r = head_fwd(*a, **kw); mark(2); return r
    model.proposal_generator.rpn_head.forward = head

    def one():
        images, u8 = model._stack_images(data)
        sizes = [tuple(s) for s in images.image_sizes]
        gt = dt.GTBatch([d["instances"] for d in data], dev, G=32)
        meta = dt.camera_meta(model.roi_heads, [torch.as_tensor(d["K"]) for d in data], [1.0] * B, sizes, dev)
        mark(0)
        x = ops.preprocess(u8, model.pixel_mean_list, model.pixel_std_list)
        features = model.backbone(x)
        if on[0]:
            features["p2"].register_hook(lambda g: (mark(11), g)[1])
        pg = model.proposal_generator
        h = pg.rpn_head([features[f] for f in pg.in_features])
        losses = dt.forward_train(model, sizes, features, h, gt, meta)
        loss = sum(losses.values())
        opt.zero_grad()
        mark(10)
        loss.backward()
        mark(12)
        opt.collect_grads(); opt.step()
        mark(13)
        torch.cuda.synchronize()

    with d2.EventStorage(0):
        one(); one()
        on[0] = True
        for k in range(len(REGIONS)):       # calibration: marker k -> grid size
            mark(k)
        torch.cuda.synchronize()
        one()


def parse(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    fills = [i for i, r in enumerate(rows) if "FillFunctor" in r["Kernel_Name"]]
    # calibration = first run of len(REGIONS) consecutive fills with strictly increasing grid
    gkey = lambda r: (int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))
    cal = None
    for j in range(len(fills) - len(REGIONS) + 1):
        idx = fills[j:j + len(REGIONS)]
        if idx[-1] - idx[0] == len(REGIONS) - 1:
            g = [gkey(rows[i])[0] for i in idx]
            if all(b > a for a, b in zip(g, g[1:])) and g[0] > 2000:
                cal = idx
    assert cal is not None, "calibration run not found"
    g2k = {gkey(rows[i]): k for k, i in enumerate(cal)}
    cur, acc = None, collections.OrderedDict()
    names = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in rows[cal[-1] + 1:]:
        if "FillFunctor" in r["Kernel_Name"] and gkey(r) in g2k:
            cur = g2k[gkey(r)]
            continue
        if cur is None or cur == len(REGIONS) - 1:
            continue
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = acc.setdefault(cur, [0, 0.0]); a[0] += 1; a[1] += d
        n = names[cur][r["Kernel_Name"][:90]]; n[0] += 1; n[1] += d
    tot = sum(v[1] for v in acc.values())
    print(f"total {tot/1e3:.3f} ms over {sum(v[0] for v in acc.values())} kernels")
    for k, (c, t) in acc.items():
        print(f"\n== {REGIONS[k]:32s} {c:5d} kernels {t/1e3:8.3f} ms")
        for nm, (cc, tt) in sorted(names[k].items(), key=lambda kv: -kv[1][1])[:22]:
            print(f"      {tt/1e3:7.3f} ms {cc:4d}x  {nm}")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else parse(sys.argv[2])
