"""attention kernel alone (ViT-L shape: 4 x 16 heads x 1370 tokens) for rocprofv3 --pmc passes"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("3dod_amd.hipops")
dev = torch.device("cuda:0")
B, N, H, D = 4, 1370, 16, 64
qkv = torch.randn(B * N, 3 * H * D, device=dev).to(torch.bfloat16)
for _ in range(10):
    ops.attention(qkv, B, N, H, D, D ** -0.5)
torch.cuda.synchronize()
