"""How the RoIs of a supervised train step (bench setup: 4 x 512^2, 512 sampled RoIs per image) spread over the 16 x 16 tiles
of the tile-owner RoIAlign backward: RoIs per (tile), and a list-scheduling estimate of the kernel's makespan on 512 resident
workgroups for the tile order it uses (level, image, y, x) and for longest-first."""
import importlib, os, sys, heapq
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CR_GRAPHS", "none")
import numpy as np, torch
bt = importlib.import_module("bench_train")
d2 = importlib.import_module("3dod_amd.d2lite")
ops = importlib.import_module("3dod_amd.hipops")
dev = torch.device("cuda:0")
cfg, model, opt, syn, solver = bt.build(dev)
batches = [syn.make_batch(4, 777 + i) for i in range(4)]
for b in batches:
    for d in b:
        for k in ("image", "instances"):
            d[k] = d[k].to(dev)
step = solver.TrainStep(cfg, model, opt, world_size=1)
seen = []
orig = ops.roi_align_pyramid
def spy(feats, rois, scales, out_size):
    seen.append((rois.detach().cpu().numpy().copy(), [tuple(f.shape) for f in feats], tuple(scales)))
    return orig(feats, rois, scales, out_size)
ops.roi_align_pyramid = spy
with d2.EventStorage(1):
    for i in range(6):
        seen.clear()
        step(batches[i % 4])
torch.cuda.synchronize()
for rois, shapes, scales in seen[:1]:
    R = len(rois)
    area = (rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2])
    minl = int(round(-np.log2(scales[0])))
    lv = np.clip(np.floor(4 + np.log2(np.sqrt(area) / 224 + 1e-8)), minl, minl + len(scales) - 1).astype(int) - minl
    print("RoIs", R, "per level", np.bincount(lv, minlength=len(scales)).tolist(), "maps", [s[1:3] for s in shapes])
    counts = []
    for l, (shape, sc) in enumerate(zip(shapes, scales)):
        N, H, W, C = shape
        ty, tx = (H + 15) // 16, (W + 15) // 16
        cnt = np.zeros((N, ty, tx), int)
        for r in rois[lv == l]:
            n = int(r[0])
            x0 = max(int(np.floor(r[1] * sc - 0.5)) - 1, 0); x1 = min(int(np.ceil(r[3] * sc - 0.5)) + 1, W - 1)
            y0 = max(int(np.floor(r[2] * sc - 0.5)) - 1, 0); y1 = min(int(np.ceil(r[4] * sc - 0.5)) + 1, H - 1)
            cnt[n, y0 // 16:y1 // 16 + 1, x0 // 16:x1 // 16 + 1] += 1
        counts.append(cnt.reshape(-1))
        print(f" level {l}: {cnt.size} tiles, RoIs per tile mean {cnt.mean():.1f} max {cnt.max()} p90 {np.percentile(cnt, 90):.0f}; incidences {cnt.sum()}")
    c = np.concatenate(counts)
    blocks = np.repeat(c, C // 64)                       # one workgroup per (tile, 64 channels)
    for a, b in ((8.0, 1.6), (8.0, 1.0)):                # us per workgroup = a + b * RoIs (a: scan of 2048 extents + tile store)
        t = a + b * blocks
        def makespan(order, slots=512):
            h = [0.0] * slots
            heapq.heapify(h)
            for i in order:
                heapq.heappush(h, heapq.heappop(h) + t[i])
            return max(h)
        print(f" model {a} + {b} * n us: sum/slots {t.sum() / 512:.0f} us, longest workgroup {t.max():.0f} us, "
              f"makespan in tile order {makespan(range(len(t))):.0f} us, longest-first {makespan(np.argsort(-t)):.0f} us")
