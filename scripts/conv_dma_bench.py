"""time per launch of cr_conv2d_fwd / cr_conv2d_bwd_data on the trunk's layer shapes under the current CR_CONV_DMA* settings
(run once per setting and compare): python scripts/conv_dma_bench.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("3dod_amd.hipops")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("CR_CONV_DMA")) or "default"
out = []
for (N, H, W, Cin, Cout, k) in [(4, 128, 128, 256, 256, 3), (4, 128, 128, 64, 64, 3), (4, 64, 64, 128, 128, 3), (4, 64, 64, 256, 256, 3),
                                (4, 64, 64, 128, 256, 1), (4, 32, 32, 256, 256, 3), (4, 32, 32, 512, 256, 1), (4, 16, 16, 512, 512, 3)]:
    x = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(Cout, k * k * Cin, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    def T(f, n=40):
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t = T(lambda: ops.conv_fwd_raw(x, w, Cout, k, 1, k // 2, relu=True))
    out.append(f"{H}^2 {Cin}->{Cout} k{k}: {t:6.1f}us")
print(tag.ljust(46), " | ".join(out))
