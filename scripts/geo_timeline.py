import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes
import bench
import numpy as np
_lib = importlib.import_module("3dod_amd._lib")
dev = "cuda:0"
inp = bench.geometry_inputs(1024, 1000, 1234, dev)
a = (inp["cubes"], inp["K"], inp["im_wh"], inp["ref"], inp["mu"], inp["sg"], inp["rect"])
ALL = ("corners", "boxes", "iou", "dim", "corner", "combined")
lib = _lib.load()
for want in (ALL, ()):
    tl = torch.zeros(1024 * 16, dtype=torch.int64, device=dev)
    N, Pn = 1024, 1000
    out = {k: torch.empty(s, device=dev) for k, s in {"corners": (N, Pn, 8, 2), "boxes": (N, Pn, 4), "iou": (N, Pn), "dim": (N, Pn), "corner": (N, Pn), "combined": (N, Pn)}.items() if k in want}
    am = torch.empty(N, dtype=torch.int64, device=dev); best = torch.empty(N, device=dev)
    def run():
        rc = lib.cr_cubes_project_score_fast(_lib.ctx_for(torch.device(dev)), _lib.ptr(a[0]), N, Pn, _lib.ptr(a[1]), 0, 512.0, 512.0, _lib.ptr(a[3]), _lib.ptr(a[4]),
            _lib.ptr(a[5]), _lib.ptr(a[6]), _lib.ptr(out.get("corners")), _lib.ptr(out.get("boxes")), _lib.ptr(out.get("iou")), _lib.ptr(out.get("dim")),
            _lib.ptr(out.get("corner")), _lib.ptr(out.get("combined")), _lib.ptr(am), _lib.ptr(best), _lib.ptr(None), _lib.ptr(tl))
        assert rc == 0
    for _ in range(5): run()
    torch.cuda.synchronize()
    t = tl.cpu().numpy().reshape(1024, 16).astype(np.int64)
    t0 = t[:, 0].min()
    print("want =", "all" if want else "none")
    for i, n in enumerate(["start", "passA done", "end"]):
        c = (t[:, i] - t0) / 100
        print(f"  {n:12s} mean {c.mean():6.2f} min {c.min():6.2f} max {c.max():6.2f}  p90 {np.percentile(c, 90):6.2f}")
    end = (t[:, 2] - t0) / 100
    xcc = t[:, 14] & 0xf
    hw = t[:, 15]
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
    print("  XCC_ID of block b for b<16:", xcc[:16].tolist())
    for x in range(8):
        m = xcc == x
        print(f"  xcc {x}: {m.sum():4d} blocks, end mean {end[m].mean():6.2f} max {end[m].max():6.2f}")
    key = xcc * 1000 + se * 100 + sh * 20 + cu
    uniq, cnt = np.unique(key, return_counts=True)
    print("  distinct CUs used:", len(uniq), "blocks per CU histogram:", np.bincount(cnt).tolist())
    slow = np.argsort(end)[-12:]
    print("  slowest blocks:", [(int(b), float(end[b]), int(key[b])) for b in slow])
