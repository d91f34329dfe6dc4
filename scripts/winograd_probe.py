"""Winograd route (ops.wino_conv3x3_group) against the direct grouped launch on the pyramid of 4 x 512 x 512 images:
forward and backward-data, results and time"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CR_WINOGRAD"] = "1"
import torch
ops = importlib.import_module("3dod_amd.hipops")
dev = "cuda:0"
torch.manual_seed(0)
sizes = [128, 64, 32, 16, 8]
C = O = 256
xs = [torch.randn(4, s, s, C, device=dev) * 0.7 for s in sizes]
w = (torch.randn(O, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5).contiguous(memory_format=torch.channels_last)
b = torch.randn(O, device=dev) * 0.1
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
# forward
yd = [torch.empty(4, s, s, O, device=dev) for s in sizes]
yw = [torch.empty(4, s, s, O, device=dev) for s in sizes]
direct = lambda: ops.conv_fwd_group_raw(xs, [w] * 5, yd, C, O, 3, 1, [b] * 5, True)
wino = lambda: ops.wino_conv3x3_group(xs, w, yw, b, True, None, False)
direct(); wino(); torch.cuda.synchronize()
ref = [torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).relu().permute(0, 2, 3, 1) for x in xs]
for name, ys in (("direct", yd), ("winograd", yw)):
    print(name, "forward max err / max:", max(float((y.double() - r).abs().max() / r.abs().max()) for y, r in zip(ys, ref)))
print(f"forward: direct {t(direct):.0f} us, winograd {t(wino):.0f} us")
# backward-data
gs = [torch.randn(4, s, s, O, device=dev) for s in sizes]
accs = [torch.randn(4, s, s, C, device=dev) for s in sizes]
wt = ops.prepared_weights(w, True, torch.float32)[1]
dd = [torch.empty(4, s, s, C, device=dev) for s in sizes]
dw_ = [torch.empty(4, s, s, C, device=dev) for s in sizes]
directb = lambda: ops.conv_bwd_data_group_raw(gs, [wt] * 5, dd, [x.shape for x in xs], C, O, 3, 1, accs)
winob = lambda: ops.wino_conv3x3_group(gs, w, dw_, None, False, accs, True)
directb(); winob(); torch.cuda.synchronize()
refb = [torch.nn.functional.conv_transpose2d(g.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1) + a.double()
        for g, a in zip(gs, accs)]
for name, ys in (("direct", dd), ("winograd", dw_)):
    print(name, "backward-data max err / max:", max(float((y.double() - r).abs().max() / r.abs().max()) for y, r in zip(ys, refb)))
print(f"backward-data: direct {t(directb):.0f} us, winograd {t(winob):.0f} us")
# weight gradient
sink_d = torch.zeros(O, C, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
sink_w = torch.zeros(O, C, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
bd_, bw_ = torch.zeros(O, device=dev), torch.zeros(O, device=dev)
directw = lambda: ops.conv_bwd_weight_group_raw(gs, xs, [sink_d] * 5, [bd_] * 5, C, O, 3, 1)
winow = lambda: ops.wino_wgrad_group(gs, xs, sink_w, bw_)
directw(); winow(); torch.cuda.synchronize()
refw = sum(torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), w.shape, g.permute(0, 3, 1, 2).double(), padding=1) for x, g in zip(xs, gs))
refb = sum(g.double().sum((0, 1, 2)) for g in gs)
print("weight gradient max err / max: direct", float((sink_d.double() - refw).abs().max() / refw.abs().max()),
      "winograd", float((sink_w.double() - refw).abs().max() / refw.abs().max()),
      "| bias: direct", float((bd_.double() - refb).abs().max() / refb.abs().max()), "winograd", float((bw_.double() - refb).abs().max() / refb.abs().max()))
print(f"weight gradient: direct {t(directw):.0f} us, winograd {t(winow):.0f} us")
