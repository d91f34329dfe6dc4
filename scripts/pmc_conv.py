"""launches the three conv directions of the roofline shape (3x3 256->256 on 4x128x128) a few times; run under
rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) to read the memory-side traffic per launch."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("3dod_amd.hipops")
dev = torch.device("cuda:0")
xs, Cout, k = (4, 128, 128, 256), 256, 3
dt = ops.act_dtype()                  # CR_PRECISION=fp32 (default) | bf16
x = torch.randn(xs, device=dev).to(dt)
w = torch.randn(Cout, xs[3], k, k, device=dev).contiguous(memory_format=torch.channels_last)
wb, wt = ops.prepared_weights(w, True, dt)
y = ops.conv_fwd_raw(x, wb, Cout, k, 1, 1)
dy = torch.randn_like(y.float()).to(dt)
sink = torch.zeros(Cout * xs[3] * k * k, device=dev)
for _ in range(5):
    ops.conv_fwd_raw(x, wb, Cout, k, 1, 1)
    ops.conv_bwd_data_raw(dy, wt, xs, k, 1, 1)
    ops.conv_bwd_weight_raw(dy, x, k, 1, 1, sink=sink)
torch.cuda.synchronize()
