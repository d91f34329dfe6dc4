"""Cubes container -- ProposalNetwork/utils/spaces.py:95-328 of the reference, same tensor layout
(N,P,15) = [cx,cy,cz,w,h,l,R row-major]; corner / projection methods run on the HIP kernels."""
import numpy as np
import torch

from ... import geometry as geo


class Cubes:
    def __init__(self, tensor, scores=None, labels=None):
        if scores is not None:
            assert scores.ndim == 2, f"scores.shape must be (n_instances, n_proposals), but was {scores.shape}"
        self.scores = scores
        self.labels = labels
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(np.asarray(tensor), dtype=torch.float32, device=torch.device("cpu"))
        else:
            tensor = tensor.to(torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 15)).to(dtype=torch.float32)
        self.tensor = tensor
        if self.tensor.dim() == 1:
            self.tensor = self.tensor.unsqueeze(0)
        if self.tensor.dim() == 2:
            self.tensor = self.tensor.unsqueeze(0)

    @property
    def centers(self):
        return self.tensor[:, :, :3]

    @property
    def dimensions(self):
        return self.tensor[:, :, 3:6]

    @property
    def rotations(self):
        shape = self.tensor.shape
        return self.tensor[:, :, 6:].reshape(shape[0], shape[1], 3, 3)

    @property
    def device(self):
        return self.tensor.device

    @property
    def num_instances(self):
        return self.tensor.shape[0]

    @property
    def shape(self):
        return self.tensor.shape

    def clone(self):
        return Cubes(self.tensor.clone())

    def get_all_corners(self):
        """(N,P,8,3) camera-space corners (spaces.py:192-204) -- cr_cuboid_corners."""
        N, P = self.tensor.shape[:2]
        t = self.tensor.reshape(N * P, 15)
        return geo.cuboid_corners(t[:, :6].contiguous(), t[:, 6:].reshape(-1, 3, 3).contiguous()).view(N, P, 8, 3)

    def _project(self, K, clamp, want):
        N = self.num_instances
        dev = self.device
        dummy3 = torch.ones((N, 3), device=dev)
        ref = torch.tensor([[0., 0., 1., 1.]], device=dev).repeat(N, 1)
        if clamp is None:
            raise UnboundLocalError("get_bube_corners(K) without `clamp` fails in the reference too (spaces.py:243)")
        return geo.cubes_project_score(self.tensor.contiguous(), K.to(dev), clamp, ref, dummy3, dummy3, None, want=want)

    def get_bube_corners(self, K, clamp: tuple = None):
        """projected, clamped corners (N,P,8,2) (spaces.py:224-245)."""
        return self._project(K, clamp, ("corners",))["corners"]

    def __len__(self):
        return self.tensor.shape[0]

    def __repr__(self):
        return f'Cubes({self.tensor})'

    def to(self, device):
        if isinstance(self.scores, torch.Tensor):
            self.scores = self.scores.to(device=device)
        if isinstance(self.labels, torch.Tensor):
            self.labels = self.labels.to(device=device)
        return Cubes(self.tensor.to(device=device), self.scores, self.labels)

    def __getitem__(self, item):
        if isinstance(item, int):
            prev_n_prop = self.tensor.shape[1]
            return Cubes(self.tensor[item].view(1, prev_n_prop, -1))
        elif isinstance(item, tuple):
            return Cubes(self.tensor[item[0], item[1]].view(1, 1, -1))
        b = self.tensor[item]
        assert b.dim() == 2, "Indexing on Cubes with {} failed to return a matrix!".format(item)
        return Cubes(b)

    @classmethod
    def cat(cls, cubes_list):
        assert isinstance(cubes_list, (list, tuple))
        if len(cubes_list) == 0:
            return cls(torch.empty(0))
        return cls(torch.cat([b.tensor for b in cubes_list], dim=0))

    def __iter__(self):
        yield from self.tensor

    def split(self, split_size, dim=1):
        return tuple(Cubes(x) for x in self.tensor.split(split_size, dim=dim))

    def reshape(self, *args):
        return Cubes(self.tensor.reshape(*args), self.scores, self.labels)
