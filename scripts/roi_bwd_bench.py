"""Times RoIAlign backward (7x7, 256 channels, 4 x 512 RoIs over the p2..p5 maps of a 512 x 512 batch) on proposal-like boxes
(clusters around a few objects per image, log-uniform sizes) and checks it against the plain per-sample kernel."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("3dod_amd.hipops")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, S, C = 4, 512, 256
rois = []
for b in range(B):
    ctr = torch.rand(6, 2, generator=g) * 400 + 56
    which = torch.randint(0, 6, (S,), generator=g)
    size = torch.exp(torch.rand(S, 2, generator=g) * 2.5 + 3.0)          # 20 .. 245 px
    c = ctr[which] + torch.randn(S, 2, generator=g) * 12
    x1y1 = (c - size / 2).clamp(0, 511); x2y2 = (c + size / 2).clamp(1, 512)
    rois.append(torch.cat([torch.full((S, 1), float(b)), x1y1, x2y2], 1))
rois = torch.cat(rois)
if os.environ.get("ROI_SORT"):       # locality experiment: RoIs of an image in (size class, y, x) order
    area = (rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2])
    lv = torch.floor(4 + torch.log2(area.sqrt() / 224 + 1e-8)).clamp(2, 5)
    cy = ((rois[:, 2] + rois[:, 4]) / 2 / 32).floor(); cx = (rois[:, 1] + rois[:, 3]) / 2
    key = ((rois[:, 0] * 4 + (lv - 2)) * 16 + cy) * 512 + cx
    rois = rois[key.argsort()]
rois = rois.to(dev)
dt = ops.act_dtype()
feats = [torch.randn(B, 512 // s, 512 // s, C, device=dev).to(dt).requires_grad_(True) for s in (4, 8, 16, 32)]
scales = (0.25, 0.125, 0.0625, 0.03125)
out = ops.roi_align_pyramid(feats, rois, scales, 7)
dout = torch.randn_like(out)

def run():
    return torch.autograd.grad(out, feats, dout, retain_graph=True)

def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

t = timeit(run)
gs = run()
os.environ["CR_ROI_BWD_PLAIN"] = "1"
ref = run()
del os.environ["CR_ROI_BWD_PLAIN"]
err = max(float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12)) for a, b in zip(gs, ref))
print(f"roi_align bwd (incl. zero fill): {t:7.1f} us   max rel err vs plain kernel {err:.2e}", flush=True)
