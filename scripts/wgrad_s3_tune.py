"""sweep of the split-mode weight-gradient kernel's knobs (one subprocess per setting: the knobs are read once per process)"""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    ops = importlib.import_module("3dod_amd.hipops")
    ops.set_precision("fp32x3")
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
for name, env in [("default", {}), ("no mfma", {"CR_S3_DBG": "1"}), ("no lds reads", {"CR_S3_DBG": "3"}), ("no atomics", {"CR_S3_DBG": "4"}),
                  ("nothing but loads+split+stores", {"CR_S3_DBG": "7"}),
                  ("blocks 256", {"CR_WG_S3_BLOCKS": "256"}), ("blocks 768", {"CR_WG_S3_BLOCKS": "768"}), ("blocks 1024", {"CR_WG_S3_BLOCKS": "1024"}),
                  ("no xcd", {"CR_XCD": "0"})]:
    r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=dict(os.environ, **env))
    print(f"{name:32s}: {r.stdout.strip()} {r.stderr.strip()[-200:] if r.returncode else ''}", flush=True)
