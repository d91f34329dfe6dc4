import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("3dod_amd.hipops")
dev = torch.device("cuda:0")
B, R, n_s, k_fg = 2, 1032, 512, 128
g = torch.Generator().manual_seed(0)
keys = torch.rand(2, B, R, generator=g).to(dev)
keys[0, :, 200:] = 0
boxes = torch.rand(B, R, 4, generator=g).to(dev)
cls = torch.randint(0, 50, (B, R), generator=g).to(dev)
am = torch.randint(0, 8, (B, R), generator=g).to(torch.int32).to(dev)
kk = min(max(k_fg, n_s), R)
tkey, tidx = keys.view(2 * B, R).topk(kk, dim=1)
torch.cuda.synchronize(); print("topk ok", tkey.shape, flush=True)
fkey, fidx = tkey[:B, :k_fg], tidx[:B, :k_fg]
bkey, bidx = tkey[B:, :n_s], tidx[B:, :n_s]
a = fidx.contiguous(); torch.cuda.synchronize(); print("contig ok", flush=True)
out = ops.roi_compact(fidx, fkey, bidx, bkey, n_s, boxes, cls, am)
torch.cuda.synchronize(); print("compact ok", out[4], flush=True)
