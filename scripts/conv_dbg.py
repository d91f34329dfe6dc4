"""times the forward igemm of a few shapes under CR_CONV_DBG ablations (set by the caller's env)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("3dod_amd.hipops")
dev = torch.device("cuda:0")
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * n) * 1e3
out = []
for xs, Cout, k in [((4, 32, 32, 256), 256, 3), ((4, 16, 16, 512), 512, 3), ((4, 64, 64, 128), 128, 3), ((4, 128, 128, 256), 256, 3)]:
    x = torch.randn(xs, device=dev).to(torch.bfloat16)
    w = torch.randn(Cout, xs[3], k, k, device=dev).contiguous(memory_format=torch.channels_last)
    wb, wt = ops.prepared_weights(w, True)
    out.append(f"{timeit(lambda: ops.conv_fwd_raw(x, wb, Cout, k, 1, 1)):7.1f}")
print(os.environ.get("CR_CONV_DBG", "0"), os.environ.get("CR_IGEMM_KU_SMALL", "4"), " ".join(out), flush=True)
