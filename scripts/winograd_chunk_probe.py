"""Winograd forward of the RPN-head pyramid: one pipeline over all maps against chunks whose V / M planes fit the Infinity
Cache (the big level split by image)"""
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
y1 = [torch.empty(4, s, s, O, device=dev) for s in sizes]
y2 = [torch.empty(4, s, s, O, device=dev) for s in sizes]
whole = lambda: ops.wino_conv3x3_group(xs, w, y1, b, True, None, False)
def chunks(per):
    groups = [([xs[0][i:i + per]], [y2[0][i:i + per]]) for i in range(0, 4, per)] + [(xs[1:], y2[1:])]
    def run():
        for a, o in groups:
            ops.wino_conv3x3_group(a, w, o, b, True, None, False)
    return run
print(f"whole {t(whole):.0f} us")
for per in (2, 1):
    f = chunks(per)
    f(); torch.cuda.synchronize()
    print(f"big level {per} image(s) per chunk: {t(f):.0f} us, equal {all(torch.equal(a, b_) for a, b_ in zip(y1, y2))}")
