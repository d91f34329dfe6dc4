"""GPU: CR_DETERMINISTIC=1 -- the f32 weight-gradient kernels write per-split partial sums to workspace slabs and a second
pass adds them in a fixed order (no float atomics); with the tile-owner RoIAlign backward, the fixed-order BatchNorm / loss
reductions and the radix top-k, a float32 train step is then bit-reproducible: two processes, same seed, three optimizer steps,
identical parameters.  (The default mode accumulates dW with f32 atomics: equal to rounding, not bitwise.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROG = r'''
import hashlib, importlib, sys, torch
sys.path.insert(0, %r)
bt = importlib.import_module("bench_train")
d2 = importlib.import_module("3dod_amd.d2lite")
dev = torch.device("cuda:0")
cfg, model, opt, syn, solver = bt.build(dev, seed=3)
step = solver.TrainStep(cfg, model, opt, world_size=1)
losses = []
with d2.EventStorage(0):
    for i in range(3):
        torch.manual_seed(100 + i)
        step(syn.make_batch(2, 40 + i))
    rep = step.report()
torch.cuda.synchronize()
print("HASH", hashlib.sha256(opt.flat_p.cpu().numpy().tobytes()).hexdigest(), rep["total_loss"], rep["iterations_explode"])
''' % ROOT


def _run(det):
    env = dict(os.environ, CR_DETERMINISTIC="1" if det else "0", CR_GRAPHS="none")
    out = subprocess.run([sys.executable, "-c", PROG], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    line = [l for l in out.stdout.splitlines() if l.startswith("HASH")]
    assert line, out.stderr[-2000:]
    _, h, loss, bad = line[-1].split()
    return h, float(loss), float(bad)


def test_deterministic_mode_is_bit_reproducible():
    a = _run(True)
    b = _run(True)
    assert a[2] == 0 and b[2] == 0
    assert a[0] == b[0], "parameters after three steps differ between two runs in deterministic mode"
    c = _run(False)
    # the default mode trains the same model with another (run-to-run varying) summation order of dW: after three SGD steps
    # the third loss agrees to a few 1e-3 (measured 1e-4 ... 1.2e-3 over runs), not to rounding
    assert abs(c[1] - a[1]) <= 1e-2 * max(1.0, abs(a[1]))
