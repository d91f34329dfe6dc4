import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bt = importlib.import_module("bench_train")
from oracle import cpu_backend
ops = importlib.import_module("3dod_amd.hipops")
dev = torch.device("cuda:0")
cfg, model, opt, syn, solver = bt.build(dev)
batch = syn.make_batch(2, 21, with_gt=False)
model.train()
fpn = model.backbone
def rel(a, b):
    a = a.float().cpu(); b = b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12)), float((a - b).abs().max() / (b.abs().max() + 1e-12))
cpu_backend.EMULATE_BF16 = True
with torch.no_grad():
    images, x = model.preprocess_image(batch)
    bu = fpn.bottom_up(x)
    prev_g = prev_c = None
    for idx, (lat, out) in enumerate(zip(fpn.lateral_convs, fpn.output_convs)):
        f = bu[fpn.in_features[-idx - 1]]
        fc = f.float().cpu()
        lg = ops.conv_bias_act(f, lat.weight, lat.bias, 1, 0)
        lc = cpu_backend.conv_bias_act(fc, lat.weight.detach().cpu(), lat.bias.detach().cpu(), 1, 0)
        print(idx, "lateral", tuple(f.shape), rel(lg, lc))
        if prev_g is None:
            pg_, pc_ = lg, lg.float().cpu()
        else:
            pg_ = ops.upsample2x_add(lg, prev_g)
            pc_ = cpu_backend.upsample2x_add(lg.float().cpu(), prev_g.float().cpu())
            print(idx, "upsample_add", rel(pg_, pc_))
        og = ops.conv_bias_act(pg_, out.weight, out.bias, 1, 1)
        oc = cpu_backend.conv_bias_act(pg_.float().cpu(), out.weight.detach().cpu(), out.bias.detach().cpu(), 1, 1)
        print(idx, "output3x3", tuple(pg_.shape), rel(og, oc))
        prev_g = pg_
