"""cr_conv2d_fwd_dma (LDS-DMA, double-buffered 64-deep stages) against cr_conv2d_fwd: equality and time per launch"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_lib = importlib.import_module("3dod_amd._lib")
ops = importlib.import_module("3dod_amd.hipops")
lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (N, H, W, Cin, Cout, k, stride) in [(4, 128, 128, 256, 256, 3, 1), (4, 64, 64, 256, 256, 3, 1), (4, 32, 32, 256, 256, 3, 1),
                                        (4, 128, 128, 64, 128, 3, 1), (4, 64, 64, 128, 256, 3, 2), (4, 128, 128, 256, 256, 1, 1),
                                        (2, 50, 37, 64, 128, 3, 1)]:
    pad = k // 2
    x = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(Cout, k * k * Cin, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y0 = ops.conv_fwd_raw(x, w, Cout, k, stride, pad, bias=b, relu=True)
    y1 = torch.empty_like(y0)
    call = lambda: _lib.check(lib.cr_conv2d_fwd(_lib.ctx_for(dev), _lib.ptr(x), _lib.ptr(w), _lib.ptr(y1), N, H, W, Cin, Cout, k,
                                                    stride, pad, _lib.ptr(b), None, 1, None, 0), "cr_conv2d_fwd")
    call()
    torch.cuda.synchronize()
    same = torch.equal(y0, y1)
    err = float((y0.float() - y1.float()).abs().max())

    def T(f, n=50):
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t0 = T(lambda: ops.conv_fwd_raw(x, w, Cout, k, stride, pad, bias=b, relu=True))
    t1 = T(call)
    fl = 2.0 * N * Ho * Wo * Cout * k * k * Cin
    print(f"{N}x{H}x{W} {Cin}->{Cout} k{k} s{stride}: equal={same} maxdiff={err:.3g}  igemm {t0:7.1f} us ({fl / t0 / 1e6:6.0f} TF/s)   dma {t1:7.1f} us ({fl / t1 / 1e6:6.0f} TF/s)")
