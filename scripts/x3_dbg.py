import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_lib = importlib.import_module("3dod_amd._lib")
g = torch.Generator().manual_seed(5)
rows, K = 48, 96
w = torch.randn(rows, K, generator=g) * torch.exp(torch.randn(rows, K, generator=g) * 8.0)
w[1, :4] = torch.tensor([3.0e-38, 1.1754944e-38, 2.0 ** 100, -(2.0 ** -100)])
w[2, :3] = torch.tensor([16777215.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24])
wd = w.cuda()
out = torch.empty(rows * K * 3, dtype=torch.bfloat16, device="cuda")
_lib.check(_lib.load().cr_weight_split3(_lib.ctx_for(wd.device), _lib.ptr(wd), _lib.ptr(out), rows, K), "x")
torch.cuda.synchronize()
pl = out.view(rows, K // 32, 3, 4, 8).double().cpu()
total = pl.sum(2)
kk = torch.empty(4, 8, dtype=torch.long)
for c in range(4):
    for e in range(8):
        kk[c, e] = 4 * c + e if e < 4 else 16 + 4 * c + e - 4
want = w.double().view(rows, K // 32, 32)[:, :, kk]
bad = (total != want).nonzero()
print("mismatches", bad.shape[0], "of", want.numel())
for idx in bad[:12]:
    i = tuple(idx.tolist())
    print(i, want[i].item(), total[i].item(), pl[i[0], i[1], :, i[2], i[3]].tolist())
