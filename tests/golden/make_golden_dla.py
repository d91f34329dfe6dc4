"""Golden vectors for the DLA-34 trunk: the REFERENCE's own DLA class
(cubercnn/modeling/backbone/dla.py:233-321, pure torch.nn) run on CPU in training mode (batch statistics)
with seeded random-init weights.  The product builds the same module tree with the same init order, so the
same seed gives the same weights; the fixture stores the input, the level outputs and per-tensor weight
checksums (not the 15 M weights).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_dla.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402

_refimport.install()
spec = importlib.util.spec_from_file_location("_ref_dla", os.path.join(_refimport.REFERENCE, "cubercnn/modeling/backbone/dla.py"))
ref_dla = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_dla)
torch.set_num_threads(4)

SEED = 1234
torch.manual_seed(SEED)
net = ref_dla.dla34(pretrained=False)
net.train()
g = torch.Generator().manual_seed(99)
x = torch.randn(2, 3, 64, 96, generator=g)
with torch.no_grad():
    b = net.base_layer(x)
    l0 = net.level0(b)
    l1 = net.level1(l0)
    l2 = net.level2(l1)
    l3 = net.level3(l2)
    l4 = net.level4(l3)
    l5 = net.level5(l4)
sd = net.state_dict()
names = [k for k in sd if not k.startswith("fc") and "num_batches" not in k and "running" not in k]
np.savez_compressed(os.path.join(HERE, "dla34_trunk.npz"), seed=np.int64(SEED), x=x.numpy(), base=b.numpy(), level1=l1.numpy(),
                    p2=l2.numpy(), p3=l3.numpy(), p4=l4.numpy(), p5=l5.numpy(),
                    weight_names=np.array(names), weight_sums=np.array([float(sd[k].double().sum()) for k in names]),
                    weight_abs=np.array([float(sd[k].double().abs().sum()) for k in names]),
                    notes="reference DLA (dla.py:233-321) forward in train mode, torch.manual_seed(1234) init; pinned by reference")
print({k: tuple(v.shape) for k, v in dict(p2=l2, p3=l3, p4=l4, p5=l5).items()}, len(names))
