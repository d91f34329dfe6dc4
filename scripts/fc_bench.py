"""Times the RoI heads' FC layers (box head fc1 2048 x 12544 -> 1024, cube head fc1 on ~512 foreground rows, fc2 1024 -> 1024)
in the three directions; environment switches (CR_SPLITK_TARGET, CR_CONV_BM64, CR_WG_F32_TM ...) are read by the library."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("3dod_amd.hipops")
dev = torch.device("cuda:0")

def timeit(f, n=10):
    for _ in range(2): f()
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

dt = ops.act_dtype()
for (R, K, O) in [(2048, 12544, 1024), (512, 12544, 1024), (2048, 1024, 1024), (512, 1024, 1024)]:
    x = torch.randn(R, K, device=dev).to(dt)
    w = (torch.randn(O, K, device=dev) * 0.01).to(dt)
    wt = w.t().contiguous()
    b = torch.zeros(O, device=dev)
    dy = torch.randn(R, O, device=dev).to(dt)
    dw = torch.zeros(O, K, device=dev); db = torch.zeros(O, device=dev)
    gf = 2.0 * R * K * O / 1e9
    tf = timeit(lambda: ops.linear_fwd_raw(x, w, b, relu=True))
    tb = timeit(lambda: ops.linear_bwd_data_raw(dy, wt))
    tw = timeit(lambda: ops.linear_bwd_weight_raw(dy, x, dw, db, True))
    print(f"{(R,K,O)}: {gf:6.1f} GF | fwd {tf:7.1f} us {gf/tf*1e3:5.0f} TF | bwdD {tb:7.1f} us {gf/tb*1e3:5.0f} TF | wgrad {tw:7.1f} us {gf/tw*1e3:5.0f} TF", flush=True)
