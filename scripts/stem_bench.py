"""Times the three stem-layer weight gradients (batch 4, 512 x 512) and checks them against the im2col kernel
(CR_WG_PATCH=0 in a second process gives the old numbers)."""
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

for (N, H, W, Cin, Cout, k, st, pd) in [(4, 512, 512, 4, 16, 7, 1, 3), (4, 512, 512, 16, 16, 3, 1, 1), (4, 512, 512, 16, 32, 3, 2, 1)]:
    dt = ops.act_dtype()
    x = torch.randn(N, H, W, Cin, device=dev).to(dt)
    Ho = (H + 2 * pd - k) // st + 1
    dy = torch.randn(N, Ho, Ho, Cout, device=dev).to(dt)
    sink = torch.zeros(Cout * Cin * k * k, device=dev)
    w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.1).contiguous(memory_format=torch.channels_last)
    wb, wt = ops.prepared_weights(w, True, dt)
    tf = timeit(lambda: ops.conv_fwd_raw(x, wb, Cout, k, st, pd))
    yk = ops.conv_fwd_raw(x, wb, Cout, k, st, pd)
    yr = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w, None, st, pd).permute(0, 2, 3, 1)
    ef = float((yk.float() - yr).abs().max() / yr.abs().max())
    tb, eb = float("nan"), float("nan")
    if Cin >= 16:
        tb = timeit(lambda: ops.conv_bwd_data_raw(dy, wt, tuple(x.shape), k, st, pd))
        dxk = ops.conv_bwd_data_raw(dy, wt, tuple(x.shape), k, st, pd)
        dxr = torch.nn.grad.conv2d_input((N, Cin, H, W), w, dy.float().permute(0, 3, 1, 2), stride=st, padding=pd).permute(0, 2, 3, 1)
        eb = float((dxk.float() - dxr).abs().max() / dxr.abs().max())
    print(f"{(N,H,W,Cin,Cout,k,st)}: fwd {tf:7.1f} us relerr {ef:.1e} | bwd-data {tb:7.1f} us relerr {eb:.1e}", flush=True)
    tw = timeit(lambda: ops.conv_bwd_weight_raw(dy, x, k, st, pd, sink=sink))
    sink.zero_()
    ops.conv_bwd_weight_raw(dy, x, k, st, pd, sink=sink)
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (Cout, Cin, k, k), dy.float().permute(0, 3, 1, 2), stride=st, padding=pd)
    ref = ref.permute(0, 2, 3, 1).reshape(-1)
    err = float((sink - ref).abs().max() / ref.abs().max())
    gf = 2.0 * dy.numel() * Cin * k * k / 1e9
    print(f"{(N,H,W,Cin,Cout,k,st)}: wgrad {tw:7.1f} us {gf/tw*1e3:5.0f} TF  relerr {err:.2e}", flush=True)
