"""Boxes / Instances / ImageList / ShapeSpec stand-ins (detectron2.structures, detectron2.layers)
[third-party behaviour restated]."""
import itertools
from collections import namedtuple
from typing import Any, Dict, List, Tuple

import torch


def cat(tensors, dim=0):
    assert isinstance(tensors, (list, tuple))
    if len(tensors) == 1:
        return tensors[0]
    return torch.cat(tensors, dim)


class ShapeSpec(namedtuple("_ShapeSpec", ["channels", "height", "width", "stride"])):
    def __new__(cls, channels=None, height=None, width=None, stride=None):
        return super().__new__(cls, channels, height, width, stride)


class Boxes:
    """(N,4) XYXY float boxes."""

    def __init__(self, tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32, device=torch.device("cpu"))
        else:
            tensor = tensor.to(torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4)).to(dtype=torch.float32)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self):
        return Boxes(self.tensor.clone())

    def to(self, device):
        return Boxes(self.tensor.to(device=device))

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size):
        assert torch.isfinite(self.tensor).all(), "Box tensor contains infinite or NaN!"
        h, w = box_size
        x1 = self.tensor[:, 0].clamp(min=0, max=w)
        y1 = self.tensor[:, 1].clamp(min=0, max=h)
        x2 = self.tensor[:, 2].clamp(min=0, max=w)
        y2 = self.tensor[:, 3].clamp(min=0, max=h)
        self.tensor = torch.stack((x1, y1, x2, y2), dim=-1)

    def nonempty(self, threshold=0.0):
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        b = self.tensor[item]
        assert b.dim() == 2, f"Indexing on Boxes with {item} failed to return a matrix!"
        return Boxes(b)

    def __len__(self):
        return self.tensor.shape[0]

    def __repr__(self):
        return "Boxes(" + str(self.tensor) + ")"

    def get_centers(self):
        return (self.tensor[:, :2] + self.tensor[:, 2:]) / 2

    def scale(self, scale_x, scale_y):
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    @classmethod
    def cat(cls, boxes_list):
        assert isinstance(boxes_list, (list, tuple))
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

    @property
    def device(self):
        return self.tensor.device

    def __iter__(self):
        yield from self.tensor


def pairwise_intersection(boxes1, boxes2):
    b1, b2 = boxes1.tensor, boxes2.tensor
    wh = torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])
    wh.clamp_(min=0)
    return wh.prod(dim=2)


def pairwise_iou(boxes1, boxes2):
    a1, a2 = boxes1.area(), boxes2.area()
    inter = pairwise_intersection(boxes1, boxes2)
    return torch.where(inter > 0, inter / (a1[:, None] + a2 - inter), torch.zeros(1, dtype=inter.dtype, device=inter.device))


def pairwise_ioa(boxes1, boxes2):
    a2 = boxes2.area()
    inter = pairwise_intersection(boxes1, boxes2)
    return torch.where(inter > 0, inter / a2, torch.zeros(1, dtype=inter.dtype, device=inter.device))


class Instances:
    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for k, v in kwargs.items():
            self.set(k, v)

    @classmethod
    def _from_fields(cls, image_size, fields):
        """an Instances around `fields` WITHOUT the per-field length checks of set(): for results whose fields were cut with
        one index / slice or moved as a whole (to, __getitem__, the packing loops of the heads), where the lengths agree by
        construction -- set() costs two Python-level length queries per field and these run ~10 fields x 64 images per batch"""
        ret = cls.__new__(cls)
        object.__setattr__(ret, "_image_size", image_size)
        object.__setattr__(ret, "_fields", fields)
        return ret

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            object.__setattr__(self, name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return self._fields[name]

    def set(self, name, value):
        # shape[0] for tensors: torch.Tensor.__len__ goes through Python and this runs ~10 times per image
        data_len = value.shape[0] if isinstance(value, torch.Tensor) else len(value)
        if self._fields:
            cur = len(self)
            assert cur == data_len, f"Adding a field of length {data_len} to a Instances of length {cur}"
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def remove(self, name):
        del self._fields[name]

    def get(self, name):
        return self._fields[name]

    def get_fields(self) -> Dict[str, Any]:
        return self._fields

    def to(self, *args, **kwargs):
        return Instances._from_fields(self._image_size, {k: (v.to(*args, **kwargs) if hasattr(v, "to") else v)
                                                         for k, v in self._fields.items()})

    def __getitem__(self, item):
        if type(item) == int:
            if item >= len(self) or item < -len(self):
                raise IndexError("Instances index out of range!")
            item = slice(item, None, len(self))
        return Instances._from_fields(self._image_size, {k: v[item] for k, v in self._fields.items()})

    def __len__(self):
        for v in self._fields.values():
            return v.shape[0] if isinstance(v, torch.Tensor) else v.__len__()
        raise NotImplementedError("Empty Instances does not support __len__!")

    def __iter__(self):
        raise NotImplementedError("`Instances` object is not iterable!")

    @staticmethod
    def cat(instance_lists: List["Instances"]) -> "Instances":
        assert len(instance_lists) > 0
        if len(instance_lists) == 1:
            return instance_lists[0]
        image_size = instance_lists[0].image_size
        ret = Instances(image_size)
        for k in instance_lists[0]._fields.keys():
            values = [i.get(k) for i in instance_lists]
            v0 = values[0]
            if isinstance(v0, torch.Tensor):
                values = torch.cat(values, dim=0)
            elif isinstance(v0, list):
                values = list(itertools.chain(*values))
            elif hasattr(type(v0), "cat"):
                values = type(v0).cat(values)
            else:
                raise ValueError("Unsupported type {} for concatenation".format(type(v0)))
            ret.set(k, values)
        return ret

    def __str__(self):
        s = self.__class__.__name__ + "("
        s += "num_instances={}, ".format(len(self) if self._fields else 0)
        s += "image_height={}, image_width={}, ".format(self._image_size[0], self._image_size[1])
        s += "fields=[{}])".format(", ".join((f"{k}: {v}" for k, v in self._fields.items())))
        return s

    __repr__ = __str__


class ImageList:
    def __init__(self, tensor, image_sizes):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    def __getitem__(self, idx):
        size = self.image_sizes[idx]
        return self.tensor[idx, ..., : size[0], : size[1]]

    def to(self, *args, **kwargs):
        return ImageList(self.tensor.to(*args, **kwargs), self.image_sizes)

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0, padding_constraints=None):
        assert len(tensors) > 0
        image_sizes = [(im.shape[-2], im.shape[-1]) for im in tensors]
        max_size = [max(s[0] for s in image_sizes), max(s[1] for s in image_sizes)]
        if padding_constraints is not None:
            size_divisibility = padding_constraints.get("size_divisibility", size_divisibility)
        if size_divisibility > 1:
            stride = size_divisibility
            max_size = [(m + (stride - 1)) // stride * stride for m in max_size]
        if len(tensors) == 1 and tuple(image_sizes[0]) == tuple(max_size):
            batched = tensors[0].unsqueeze(0)
        elif all(tuple(s) == tuple(max_size) for s in image_sizes):
            batched = torch.stack(list(tensors))                 # nothing to pad: one launch instead of a fill + a copy per image
        else:
            shape = [len(tensors)] + list(tensors[0].shape[:-2]) + list(max_size)
            batched = tensors[0].new_full(shape, pad_value)
            for i, img in enumerate(tensors):
                batched[i, ..., : img.shape[-2], : img.shape[-1]].copy_(img)
        return ImageList(batched.contiguous(), image_sizes)
