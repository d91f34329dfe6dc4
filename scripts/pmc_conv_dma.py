"""the roofline shape (3x3 256->256 on 4x128x128) through forward, backward-data (LDS-DMA kernel) and backward-weight for PMC
passes: rocprofv3 --pmc <counters> --kernel-trace -- python3 scripts/pmc_conv_dma.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("3dod_amd.hipops")
dev = torch.device("cuda:0")
xs, Cout = (4, 128, 128, 256), 256
x = torch.randn(xs, device=dev).to(torch.bfloat16)
w = torch.randn(Cout, xs[3], 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
wb, wt = ops.prepared_weights(w, True)
y = ops.conv_fwd_raw(x, wb, Cout, 3, 1, 1)
dy = torch.randn_like(y.float()).to(torch.bfloat16)
sink = torch.zeros(Cout * xs[3] * 9, device=dev)
for _ in range(5):
    ops.conv_fwd_raw(x, wb, Cout, 3, 1, 1)
    ops.conv_bwd_data_raw(dy, wt, x.shape, 3, 1, 1)
    ops.conv_bwd_weight_raw(dy, x, 3, 1, 1, sink=sink)
torch.cuda.synchronize()
