"""sweep of the f32 weight-gradient kernel's split count / blocks-per-CU cap on a few layer shapes (one subprocess per
setting: the knobs are read once per process).  python scripts/wgrad_f32_tune.py"""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    ops = importlib.import_module("3dod_amd.hipops")
    dev = torch.device("cuda:0")
    out = []
    for (N, H, W, Cin, Cout, k) in [(4, 128, 128, 256, 256, 3), (4, 64, 64, 256, 256, 3), (4, 32, 32, 256, 256, 3), (4, 64, 64, 128, 128, 3)]:
        x = torch.randn(N, H, W, Cin, device=dev); dy = torch.randn(N, H, W, Cout, device=dev)
        sink = torch.zeros(Cout * Cin * k * k, device=dev)
        f = lambda: ops.conv_bwd_weight_raw(dy, x, k, 1, k // 2, sink=sink)
        for _ in range(3): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): f()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        gf = 2.0 * N * H * W * Cout * k * k * Cin / 1e9
        out.append(f"{us:7.1f}us {gf / us * 1e3:5.0f}TF")
    print(" | ".join(out), flush=True)
    sys.exit(0)
for splits in (0, 7, 14, 21, 28):
    for pad in (0, 8192, 45056):
        env = dict(os.environ, CR_WG_SPLITS_F32=str(splits), CR_WG_F32_LDS_PAD=str(pad))
        r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=env)
        print(f"splits={splits:2d} lds_pad={pad:5d}: {r.stdout.strip()} {r.stderr.strip()[-200:] if r.returncode else ''}", flush=True)
