"""RoIAlign backward of the train step (4 x 512^2, 2048 RoIs, C = 256, f32): tile-owner kernel vs the separable atomic one"""
import ctypes, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
t = importlib.import_module("test_gpu_roi_bwd_tiles")
g = torch.Generator().manual_seed(0)
N, size, R, C = 4, 512, 2048, 256
shapes = [(N, size // s, size // s, C) for s in (4, 8, 16, 32, 64)]
for clustered in (True, False):
    rois = t._rois(N, R, size, g, clustered).to(t.DEV)
    dout = torch.randn(R, 7, 7, C, generator=g).to(t.DEV)
    for kind in ("tiles", "atomic"):
        for _ in range(3):
            t._run(kind, shapes, rois, dout, C)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        bufs = [torch.empty(s, dtype=torch.float32, device=t.DEV) for s in shapes]
        e0.record()
        for _ in range(10):
            if kind == "atomic":
                torch._foreach_zero_(bufs)               # the 89 MB zero fill the atomic kernel needs (what the step pays)
            t._run(kind, shapes, rois, dout, C, grads=bufs)
        e1.record(); torch.cuda.synchronize()
        print(f"clustered={clustered} {kind:7s} {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us per call")
