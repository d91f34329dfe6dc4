"""Fixture-generation helper (runs ONLY in the build container, never on the GPU
box and never from the product path): makes the reference's pure-torch geometry
importable from /root/reference by fabricating inert stub modules for the
third-party packages that are not installed, and patches
in small REAL stand-ins for the few third-party symbols that carry arithmetic.

Stand-ins are this repo's own restatement of the third-party definition, so any
golden value that flows through one of them is labelled "parity unpinned" in the
fixture's `notes` entry.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import torch

REFERENCE = os.environ.get("CR_REFERENCE", "/root/reference")
STUB = ('cv2', 'detectron2', 'pytorch3d', 'fvcore', 'iopath', 'torchvision', 'pyransac3d',
        'segment_anything', 'pycocotools', 'wandb', 'open3d', 'seaborn')


class _D:
    def __init__(s, *a, **k): pass
    def __call__(s, *a, **k):
        # used as a decorator (@REGISTRY.register(), @configurable): hand the class / function back untouched
        if len(a) == 1 and not k and (isinstance(a[0], type) or callable(a[0])) and not isinstance(a[0], _D):
            return a[0]
        return _D()
    def __getattr__(s, n):
        if n.startswith('__'):
            raise AttributeError(n)
        return _D()
    def __mro_entries__(s, b): return (object,)
    def __iter__(s): return iter(())


class _M(types.ModuleType):
    __path__ = []
    def __getattr__(s, n):
        if n.startswith('__'):
            raise AttributeError(n)
        return _D()


class _F(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(s, name, path, target=None):
        if name.split('.')[0] in STUB:
            return importlib.machinery.ModuleSpec(name, s, is_package=True)
    def create_module(s, spec): return _M(spec.name)
    def exec_module(s, m): pass


# --- real stand-ins (third-party arithmetic restated from the public definition) ---
class Boxes:
    """detectron2.structures.Boxes stand-in: XYXY float tensor (N,4)."""
    def __init__(self, tensor):
        tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4))
        self.tensor = tensor
    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    def __len__(self): return self.tensor.shape[0]
    def __getitem__(self, i):
        if isinstance(i, int):
            return Boxes(self.tensor[i].view(1, -1))
        return Boxes(self.tensor[i])
    @property
    def device(self): return self.tensor.device


def pairwise_iou(boxes1, boxes2):
    a1, a2 = boxes1.area(), boxes2.area()
    b1, b2 = boxes1.tensor, boxes2.tensor
    wh = torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])
    wh.clamp_(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (a1[:, None] + a2 - inter),
                       torch.zeros(1, dtype=inter.dtype))


def install():
    if not any(isinstance(f, _F) for f in sys.meta_path):
        sys.meta_path.insert(0, _F())
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    sys.dont_write_bytecode = True


def load():
    """returns a namespace of the reference modules used for fixtures."""
    install()
    from cubercnn.util import math_util
    from ProposalNetwork.utils import spaces, conversions, utils
    from ProposalNetwork.proposals import proposals
    from ProposalNetwork.scoring import scorefunction
    conversions.Boxes = Boxes
    utils.pairwise_iou = pairwise_iou
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "_ref_plane", os.path.join(REFERENCE, "ProposalNetwork/utils/plane.py"))
    plane = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(plane)
    ns = types.SimpleNamespace(math_util=math_util, spaces=spaces, conversions=conversions,
                               utils=utils, proposals=proposals, scorefunction=scorefunction,
                               plane=plane, Boxes=Boxes, pairwise_iou=pairwise_iou)
    return ns
