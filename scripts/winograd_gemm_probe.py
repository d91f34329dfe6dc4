"""what the matrix part of a Winograd F(2x2, 3x3) version of the grouped pyramid convolution (3x3 256 -> 256 over the five
levels of 4 x 512 x 512 images) would cost on the existing f32 GEMM kernel: 16 products (tiles x 256) @ (256 x 256), one
per position of the 4 x 4 transformed tile, against the direct grouped launch (0.83 ms).  The transforms are elementwise
passes: input 89 MB -> 358 MB, output 358 MB -> 89 MB (unfused)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("3dod_amd.hipops")
dev = "cuda:0"
tiles = 4 * sum((s // 2) ** 2 for s in (128, 64, 32, 16, 8))
print("2x2 output tiles:", tiles)
V = [torch.randn(tiles, 256, device=dev) for _ in range(16)]
U = [torch.randn(256, 256, device=dev) * 0.05 for _ in range(16)]
def run():
    return [ops.linear_fwd_raw(v, u, None) for v, u in zip(V, U)]
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
gf = 16 * 2 * tiles * 256 * 256 / 1e9
print(f"16 GEMMs: {us:.0f} us = {gf / us * 1e-3:.1f} TFLOP/s on {gf:.1f} GFLOP (direct: 103 GFLOP in 830 us)")
big = torch.randn(16 * tiles, 256, device=dev)
us1 = None
for _ in range(3): ops.linear_fwd_raw(big, U[0], None)
e0.record()
for _ in range(20): ops.linear_fwd_raw(big, U[0], None)
e1.record(); torch.cuda.synchronize()
us1 = e0.elapsed_time(e1) / 20 * 1e3
print(f"same flops as ONE launch ({16 * tiles} x 256 @ 256 x 256): {us1:.0f} us = {gf / us1 * 1e-3:.1f} TFLOP/s")
print("the transforms are elementwise passes over 89 + 358 MB (input) and 358 + 89 MB (output): >= 150 us at 6 TB/s when not fused")
# the same 16 products as TWO grouped 1x1 launches of 8 problems each (cr_conv2d_fwd_group, what a Winograd path would call)
Vs = [v.view(1, 1, tiles, 256) for v in V]
Us = [u.view(256, 256, 1, 1).contiguous(memory_format=torch.channels_last) for u in U]
Ms = [torch.empty(1, 1, tiles, 256, device=dev) for _ in range(16)]
def run2():
    for h in (0, 8):
        ops.conv_fwd_group_raw(Vs[h:h + 8], [u.view(256, 256) for u in U[h:h + 8]], Ms[h:h + 8], 256, 256, 1, 0, [None] * 8, False)
for _ in range(5): run2()
e0.record()
for _ in range(20): run2()
e1.record(); torch.cuda.synchronize()
us2 = e0.elapsed_time(e1) / 20 * 1e3
print(f"two grouped launches of 8: {us2:.0f} us = {gf / us2 * 1e-3:.1f} TFLOP/s")
ref = V[3] @ U[3].t()
print("max |grouped - matmul| / max:", float((Ms[3].view(tiles, 256) - ref).abs().max() / ref.abs().max()))
