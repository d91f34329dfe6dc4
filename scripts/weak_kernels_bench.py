"""timings of the weak-head kernels at the weak train step's sizes (2 x 512 x 512 maps, 256 foreground RoIs)"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
geo = importlib.import_module("3dod_amd.geometry")
syn = importlib.import_module("3dod_amd.synthetic")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
n, H, W = 256, 512, 512
ctr = torch.rand(n, 1, 2, generator=g) * 400 + 56
pts = (ctr + (torch.rand(n, 8, 2, generator=g) - 0.5) * 160).clamp(0, 511).to(dev)
masks = (torch.rand(32, H, W, generator=g) > 0.5).to(torch.uint8).to(dev)
midx = torch.randint(0, 32, (n,), generator=g).to(torch.int32).to(dev)
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("hull8 us", timeit(lambda: geo.hull8(pts)))
order, count, bump = geo.hull8(pts)
hull = torch.gather(pts + bump[..., None], 1, order[..., None].expand(-1, -1, 2)).requires_grad_()
print("polygon_focal fwd (+grad) us", timeit(lambda: geo.polygon_focal(hull, count, masks, midx)))
print("polygon_focal fwd only us", timeit(lambda: geo.polygon_focal(hull.detach(), count, masks, midx)))
print("pixels x rois per second (G)", n * H * W / (timeit(lambda: geo.polygon_focal(hull, count, masks, midx)) * 1e-6) / 1e9)
mask1 = masks[0]
c1k = (torch.rand(1000, 1, 2, generator=g) * 400 + 56 + (torch.rand(1000, 8, 2, generator=g) - 0.5) * 200).to(dev)
print("segment_counts (1000 proposals) us", timeit(lambda: geo.segment_counts(c1k, mask1, 4)))
