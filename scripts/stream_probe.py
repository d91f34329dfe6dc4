"""practical HBM rate for a launch of the geometry kernel's size: device copies moving 160 MB in total (read + write),
a 61 MB read-only reduction and a 98 MB fill"""
import torch
dev = "cuda:0"
def t(fn, n=200):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
src = torch.randn(20 * 1000 * 1000, device=dev)      # 80 MB
dst = torch.empty_like(src)
us = t(lambda: dst.copy_(src)); print(f"copy 80 MB -> 80 MB: {us:.1f} us, {160e6 / us / 1e6:.2f} TB/s")
a = torch.randn(61_440_000 // 4, device=dev)
us = t(lambda: a.sum()); print(f"sum over 61 MB: {us:.1f} us, {61.44e6 / us / 1e6:.2f} TB/s")
b = torch.empty(98_304_000 // 4, device=dev)
us = t(lambda: b.fill_(1.0)); print(f"fill 98 MB: {us:.1f} us, {98.3e6 / us / 1e6:.2f} TB/s")
big = torch.randn(256 * 1000 * 1000, device=dev); bd = torch.empty_like(big)
us = t(lambda: bd.copy_(big), 20); print(f"copy 1 GB -> 1 GB: {us:.1f} us, {2.048e9 / us / 1e6:.2f} TB/s")
