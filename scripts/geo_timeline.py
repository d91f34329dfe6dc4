import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes
import bench
import numpy as np
_lib = importlib.import_module("3dod_amd._lib")
geo = importlib.import_module("3dod_amd.geometry")
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
    names = ["start", "chunk0 ready", "chunk1 ready", "chunk2 ready", "chunk3 ready", "passA done", "maxc", "flags", "cand done", "argmax", "end"]
    print("want =", "all" if want else "none", "(10 ns ticks relative to first block start; mean / min / max over blocks)")
    for i, n in enumerate(names):
        c = t[:, i] - t0
        print(f"  {n:14s} {c.mean() / 100:7.2f} us  {c.min() / 100:7.2f}  {c.max() / 100:7.2f}")
