"""f32 weight-gradient split count on the mid-size layers (one subprocess per setting)"""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    ops = importlib.import_module("3dod_amd.hipops")
    dev = torch.device("cuda:0")
    out = []
    for (N, H, W, Cin, Cout, k) in [(4, 64, 64, 128, 128, 3), (4, 32, 32, 256, 256, 3), (4, 16, 16, 512, 512, 3), (4, 128, 128, 64, 64, 3), (4, 64, 64, 256, 256, 3), (4, 16, 16, 256, 256, 3), (4, 8, 8, 256, 256, 3)]:
        x = torch.randn(N, H, W, Cin, device=dev); dy = torch.randn(N, H, W, Cout, device=dev)
        sink = torch.zeros(Cout * Cin * k * k, device=dev)
        f = lambda: ops.conv_bwd_weight_raw(dy, x, k, 1, k // 2, sink=sink)
        for _ in range(3): f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): f()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3): g.replay()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 60 * 1e3
        gf = 2.0 * N * H * W * Cout * k * k * Cin / 1e9
        out.append(f"{us:6.1f}us {gf / us * 1e3:4.0f}TF")
    print(" | ".join(out), flush=True)
    sys.exit(0)
for tmcap in (128, 64):
    for splits in [int(v) for v in os.environ.get("SPLITS", "0,14,21,28,42").split(",")]:
        env = dict(os.environ, CR_WG_SPLITS_F32=str(splits), CR_WG_F32_TM=str(tmcap))
        r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=env)
        print(f"TM<={tmcap:3d} splits={splits:2d}: {r.stdout.strip()} {r.stderr.strip()[-200:] if r.returncode else ''}", flush=True)
