"""Times ops.topk on the four selections of the train step (synthetic keys of the same distributions) and checks them
against torch.topk (values exact; indices exact where the values are distinct)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("3dod_amd.hipops")
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n): f()
    gr.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): gr.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * n) * 1e3

cases = {}
iou = torch.rand(8, 65472, device=dev, generator=g) ** 6                       # mostly small IoUs, a few large
cases["rpn sampling keys (8 x 65472, k 256)"] = ((iou + 1e-6) / torch.empty_like(iou).exponential_(generator=g), 256)
sizes = [49152, 12288, 3072, 768, 192]
pad = torch.full((4, 5, 49152), float("-inf"), device=dev)
for l, s in enumerate(sizes):
    pad[:, l, :s] = torch.randn(4, s, device=dev, generator=g) * 0.02           # random-init RPN logits
cases["pre-NMS logits (20 x 49152 padded, k 2000)"] = (pad.view(20, 49152), 2000)
sc = torch.randn(4, 10000, device=dev, generator=g) * 0.02
sc[torch.rand(4, 10000, device=dev, generator=g) < 0.6] = float("-inf")        # suppressed by NMS
cases["post-NMS scores (4 x 10000, k 2000)"] = (sc, 2000)
cases["RoI sampling keys (8 x 2058, k 512)"] = ((torch.rand(8, 2058, device=dev, generator=g) + 1e-6) /
                                               torch.empty(8, 2058, device=dev).exponential_(generator=g), 512)
for name, (x, k) in cases.items():
    v, i = ops.topk(x, k)
    rv, ri = torch.topk(x, k, dim=1)
    assert torch.equal(v, rv), name
    same = (i == ri) | (torch.gather(x, 1, i) == torch.gather(x, 1, ri))
    assert bool(same.all()), name
    print(f"{name:48s} {timeit(lambda: ops.topk(x, k)):7.1f} us", flush=True)
