"""Cubes -- the proposal container of the 1000-cube method.

API and tensor layout are the reference's (ProposalNetwork/utils/spaces.py:95-328): one float32 tensor (N objects,
P proposals, 15) = [cx, cy, cz | w, h, l | R00 .. R22 row-major] plus optional per-proposal `scores` / `labels`.  The
container itself is this build's own (a thin view over the (N,P,15) tensor with a slice table); the two methods that do
arithmetic -- corners and projection -- run on the HIP geometry kernels."""
import numpy as np
import torch

from ... import geometry as geo

_FIELDS = {"centers": (slice(0, 3), None), "dimensions": (slice(3, 6), None), "rotations": (slice(6, 15), (3, 3))}


def _to_npc(t):
    """anything array-like -> float32 tensor of shape (N, P, 15): a single cube (15,) and a list of cubes (P,15) gain the
    missing leading axes, an empty input becomes (1, 0, 15)"""
    t = t.float() if torch.is_tensor(t) else torch.as_tensor(np.asarray(t), dtype=torch.float32)
    if t.numel() == 0:
        t = t.reshape(-1, 15)
    while t.dim() < 3:
        t = t[None]
    return t


class Cubes:
    def __init__(self, tensor, scores=None, labels=None):
        if scores is not None and scores.ndim != 2:
            raise AssertionError(f"scores must be (objects, proposals); got shape {tuple(scores.shape)}")
        self.tensor, self.scores, self.labels = _to_npc(tensor), scores, labels

    # ---- views ----------------------------------------------------------------------------------------------------
    def __getattr__(self, name):
        spec = _FIELDS.get(name)
        if spec is None:
            raise AttributeError(name)
        cols, unflatten = spec
        v = self.tensor[..., cols]
        return v.unflatten(-1, unflatten) if unflatten else v

    device = property(lambda self: self.tensor.device)
    shape = property(lambda self: self.tensor.shape)
    num_instances = property(lambda self: self.tensor.shape[0])

    def __len__(self):
        return self.num_instances

    def __iter__(self):
        return iter(self.tensor)

    def __repr__(self):
        return f"Cubes({self.tensor})"

    # ---- indexing / reshaping (each returns a new container, always (N,P,15)) ----------------------------------------
    def __getitem__(self, item):
        if isinstance(item, int):                        # one object, all of its proposals
            return Cubes(self.tensor[item][None])
        if isinstance(item, tuple):                      # (object, proposal) -> a single cube
            return Cubes(self.tensor[item[0], item[1]].reshape(1, 1, -1))
        picked = self.tensor[item]
        if picked.dim() != 2:
            raise AssertionError(f"index {item!r} on Cubes must select a (proposals, 15) matrix, got {tuple(picked.shape)}")
        return Cubes(picked)

    def clone(self):
        return Cubes(self.tensor.clone())

    def to(self, device):
        mv = lambda v: v.to(device=device) if torch.is_tensor(v) else v
        self.scores, self.labels = mv(self.scores), mv(self.labels)
        return Cubes(self.tensor.to(device=device), self.scores, self.labels)

    def reshape(self, *shape):
        return Cubes(self.tensor.reshape(*shape), self.scores, self.labels)

    def split(self, split_size, dim=1):
        return tuple(map(Cubes, self.tensor.split(split_size, dim=dim)))

    @classmethod
    def cat(cls, cubes_list):
        if not isinstance(cubes_list, (list, tuple)):
            raise AssertionError("Cubes.cat takes a list or tuple of Cubes")
        return cls(torch.cat([c.tensor for c in cubes_list], dim=0) if cubes_list else torch.empty(0))

    # ---- geometry (HIP kernels) --------------------------------------------------------------------------------------
    def get_all_corners(self):
        """(N,P,8,3) camera-space corners (spaces.py:192-204) -- cr_cuboid_corners."""
        N, P = self.tensor.shape[:2]
        flat = self.tensor.reshape(N * P, 15)
        return geo.cuboid_corners(flat[:, :6].contiguous(), flat[:, 6:].reshape(-1, 3, 3).contiguous()).view(N, P, 8, 3)

    def _project(self, K, clamp, want):
        if clamp is None:
            raise UnboundLocalError("get_bube_corners(K) without `clamp` fails in the reference too (spaces.py:243)")
        N, dev = self.num_instances, self.device
        ones3 = torch.ones((N, 3), device=dev)
        unit_box = torch.tensor([[0.0, 0.0, 1.0, 1.0]], device=dev).repeat(N, 1)
        return geo.cubes_project_score(self.tensor.contiguous(), K.to(dev), clamp, unit_box, ones3, ones3, None, want=want)

    def get_bube_corners(self, K, clamp: tuple = None):
        """projected, clamped corners (N,P,8,2) (spaces.py:224-245)."""
        return self._project(K, clamp, ("corners",))["corners"]
