"""Times every distinct conv shape of the DLA34-FPN-RPN train step (batch 4, 512x512) in all three directions.
Shapes are captured from one real forward pass; each kernel is timed with HIP events over 20 launches."""
import importlib, os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bt = importlib.import_module("bench_train")
ops = importlib.import_module("3dod_amd.hipops")
d2 = importlib.import_module("3dod_amd.d2lite")
dev = torch.device("cuda:0")
cfg, model, opt, syn, solver = bt.build(dev)
shapes = collections.Counter()
orig = ops.conv_fwd_raw
def spy(x, wb, Cout, k, stride, pad, **kw):
    shapes[(tuple(x.shape), Cout, k, stride, pad)] += 1
    return orig(x, wb, Cout, k, stride, pad, **kw)
ops.conv_fwd_raw = spy
data = syn.make_batch(bt.IMS_PER_GPU, 1)
for d in data:
    d["image"] = d["image"].to(dev); d["instances"] = d["instances"].to(dev)
with d2.EventStorage(0):
    model(data)
ops.conv_fwd_raw = orig
torch.cuda.synchronize()

def timeit(f, n=20):
    """n launches inside one HIP graph (no host launch gaps), replayed 3 times; us per launch."""
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
    return a.elapsed_time(b) / (3 * n) * 1e3     # us

tot = collections.Counter()
print(f"{'N,H,W,Cin':>20s} {'Cout':>4s} k s  cnt | {'GF':>6s} | {'fwd us':>7s} {'TF':>5s} | {'bwdD us':>7s} {'TF':>5s} | {'wgrad us':>8s} {'TF':>5s}")
for (xs, Cout, k, stride, pad), cnt in sorted(shapes.items(), key=lambda kv: -kv[0][0][1] * 1000 - kv[0][1]):
    N, H, W, Cin = xs
    dt = ops.act_dtype()                # CR_PRECISION=fp32 (default) | bf16
    x = torch.randn(xs, device=dev).to(dt)
    w = torch.randn(Cout, Cin, k, k, device=dev).contiguous(memory_format=torch.channels_last)
    wb, wt = ops.prepared_weights(w, True, dt)
    y = ops.conv_fwd_raw(x, wb, Cout, k, stride, pad)
    dy = torch.randn_like(y.float()).to(dt)
    sink = torch.zeros(Cout * Cin * k * k, device=dev)
    gf = 2.0 * y.numel() * Cin * k * k / 1e9
    only_wg = "--only-wg" in sys.argv
    tf = timeit(lambda: ops.conv_fwd_raw(x, wb, Cout, k, stride, pad)) if not only_wg else float("nan")
    tb = timeit(lambda: ops.conv_bwd_data_raw(dy, wt, xs, k, stride, pad)) if Cin >= 16 and not only_wg else float("nan")
    tw = timeit(lambda: ops.conv_bwd_weight_raw(dy, x, k, stride, pad, sink=sink))
    tot["fwd"] += (tf if tf == tf else 0) * cnt; tot["bwd"] += (tb if tb == tb else 0) * cnt; tot["wg"] += tw * cnt
    print(f"{str(xs):>20s} {Cout:4d} {k} {stride} {cnt:4d} | {gf:6.2f} | {tf:7.1f} {gf/tf*1e3:5.0f} | {tb:7.1f} {gf/tb*1e3:5.0f} | {tw:8.1f} {gf/tw*1e3:5.0f}", flush=True)
print("per-step totals (us):", dict(tot))
