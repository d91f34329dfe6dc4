"""Stand-ins for the detectron2 data utilities the Omni3D data path is written against (detectron2 is not installed):
catalogs, BoxMode, Keypoints, the geometric transforms used by `build_augmentation` (ResizeShortestEdge + RandomFlip),
image reading, the samplers and the batch loader.  Each restates detectron2's / fvcore's documented behaviour
[third-party, absent from the reference tree -- SURVEY.md 8c: "parity unpinned"]; the reference's own use of them is
`cubercnn/data/build.py`, `cubercnn/data/dataset_mapper.py` and `cubercnn/data/datasets.py`.
"""
import itertools
import math
import types

import numpy as np
import torch
import torch.utils.data as tud


# ------------------------------------------------------------------ catalogs
class _DatasetCatalog(dict):
    """name -> zero-argument function returning list[dict]."""

    def register(self, name, func):
        assert callable(func), "You must register a function with `DatasetCatalog.register`!"
        assert name not in self, "Dataset '{}' is already registered!".format(name)
        self[name] = func

    def get(self, name):
        try:
            f = self[name]
        except KeyError as e:
            raise KeyError("Dataset '{}' is not registered! Available datasets are: {}".format(
                name, ", ".join(self.keys()))) from e
        return f()

    def remove(self, name):
        self.pop(name)


class Metadata(types.SimpleNamespace):
    def set(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)
        return self

    def get(self, key, default=None):
        return getattr(self, key, default)

    def as_dict(self):
        return dict(self.__dict__)


class _MetadataCatalog(dict):
    def get(self, name):
        assert len(name)
        if name not in self:
            self[name] = Metadata(name=name)
        return self[name]

    def remove(self, name):
        self.pop(name)


DatasetCatalog = _DatasetCatalog()
MetadataCatalog = _MetadataCatalog()


# ------------------------------------------------------------------ boxes / keypoints
class BoxMode:
    XYXY_ABS = 0
    XYWH_ABS = 1

    @staticmethod
    def convert(box, from_mode, to_mode):
        """a single box given as list / tuple comes back as the same type; arrays and tensors (…,4) are converted
        row-wise into a new array / tensor."""
        if from_mode == to_mode:
            return box
        assert {from_mode, to_mode} == {BoxMode.XYXY_ABS, BoxMode.XYWH_ABS}, "unsupported box mode"
        single = isinstance(box, (list, tuple))
        if single:
            assert len(box) == 4
            arr = np.array(box, dtype=np.float64)[None, :]
        elif isinstance(box, np.ndarray):
            arr = box.astype(np.float64).reshape(-1, 4) if box.dtype.kind != "f" else box.copy().reshape(-1, 4)
        else:
            arr = box.clone().reshape(-1, 4)
        if to_mode == BoxMode.XYXY_ABS:
            arr[:, 2] += arr[:, 0]
            arr[:, 3] += arr[:, 1]
        else:
            arr[:, 2] -= arr[:, 0]
            arr[:, 3] -= arr[:, 1]
        if single:
            return type(box)(arr.flatten().tolist())
        return arr.reshape(box.shape)


class Keypoints:
    """(N, K, 3) float tensor of (x, y, visibility)."""

    def __init__(self, keypoints):
        device = keypoints.device if isinstance(keypoints, torch.Tensor) else torch.device("cpu")
        keypoints = torch.as_tensor(keypoints, dtype=torch.float32, device=device)
        if keypoints.numel() == 0:
            keypoints = keypoints.reshape(0, 0, 3) if keypoints.dim() != 3 else keypoints
        assert keypoints.dim() == 3 and keypoints.shape[2] == 3, keypoints.shape
        self.tensor = keypoints

    def __len__(self):
        return self.tensor.size(0)

    def to(self, *args, **kwargs):
        return type(self)(self.tensor.to(*args, **kwargs))

    @property
    def device(self):
        return self.tensor.device

    def __getitem__(self, item):
        if isinstance(item, int):
            return Keypoints(self.tensor[item][None])
        return Keypoints(self.tensor[item])

    @staticmethod
    def cat(keypoints_list):
        return Keypoints(torch.cat([k.tensor for k in keypoints_list], dim=0))


# ------------------------------------------------------------------ transforms
class Transform:
    def apply_image(self, img, interp=None):
        raise NotImplementedError

    def apply_coords(self, coords):
        raise NotImplementedError

    def apply_box(self, box):
        """box (N,4) XYXY -> axis-aligned hull of the 4 transformed corners."""
        box = np.asarray(box, dtype=np.float64)
        idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
        coords = np.asarray(box).reshape(-1, 4)[:, idxs].reshape(-1, 2)
        coords = self.apply_coords(coords).reshape((-1, 4, 2))
        minxy, maxxy = coords.min(axis=1), coords.max(axis=1)
        return np.concatenate((minxy, maxxy), axis=1)


class NoOpTransform(Transform):
    def apply_image(self, img, interp=None):
        return img

    def apply_coords(self, coords):
        return coords


class HFlipTransform(Transform):
    def __init__(self, width):
        self.width = width

    def apply_image(self, img, interp=None):
        return np.flip(img, axis=1) if img.ndim <= 3 else np.flip(img, axis=-2)

    def apply_coords(self, coords):
        coords[:, 0] = self.width - coords[:, 0]
        return coords


class VFlipTransform(Transform):
    def __init__(self, height):
        self.height = height

    def apply_image(self, img, interp=None):
        return np.flip(img, axis=0) if img.ndim <= 3 else np.flip(img, axis=-3)

    def apply_coords(self, coords):
        coords[:, 1] = self.height - coords[:, 1]
        return coords


class ResizeTransform(Transform):
    def __init__(self, h, w, new_h, new_w, interp=None):
        self.h, self.w, self.new_h, self.new_w = h, w, new_h, new_w
        self.interp = interp            # PIL resampling filter; None -> bilinear

    def apply_image(self, img, interp=None):
        assert img.shape[:2] == (self.h, self.w), (img.shape, self.h, self.w)
        from PIL import Image
        filt = interp if interp is not None else (self.interp if self.interp is not None else Image.BILINEAR)
        if img.dtype == np.uint8:
            if img.ndim > 2 and img.shape[2] == 1:
                pil = Image.fromarray(img[:, :, 0], mode="L")
            else:
                pil = Image.fromarray(img)
            ret = np.asarray(pil.resize((self.new_w, self.new_h), filt))
            if img.ndim > 2 and img.shape[2] == 1:
                ret = np.expand_dims(ret, -1)
            return ret
        # float images go through torch's interpolate, as in detectron2
        t = torch.from_numpy(np.ascontiguousarray(img))
        shape = list(t.shape)
        t = t.view(shape[0], shape[1], -1).permute(2, 0, 1)[None].float()
        mode = {Image.NEAREST: "nearest", Image.BILINEAR: "bilinear", Image.BICUBIC: "bicubic"}[filt]
        t = torch.nn.functional.interpolate(t, (self.new_h, self.new_w), mode=mode,
                                            align_corners=None if mode == "nearest" else False)
        shape[:2] = (self.new_h, self.new_w)
        return t[0].permute(1, 2, 0).reshape(shape).numpy()

    def apply_coords(self, coords):
        coords[:, 0] = coords[:, 0] * (self.new_w * 1.0 / self.w)
        coords[:, 1] = coords[:, 1] * (self.new_h * 1.0 / self.h)
        return coords


class TransformList(Transform):
    def __init__(self, transforms):
        flat = []
        for t in transforms:
            flat.extend(t.transforms if isinstance(t, TransformList) else [t])
        self.transforms = flat

    def apply_image(self, img, interp=None):
        for t in self.transforms:
            img = t.apply_image(img)
        return img

    def apply_coords(self, coords):
        for t in self.transforms:
            coords = t.apply_coords(coords)
        return coords

    def apply_box(self, box):
        for t in self.transforms:
            box = t.apply_box(box)
        return box

    def __iter__(self):
        return iter(self.transforms)

    def __len__(self):
        return len(self.transforms)

    def __getitem__(self, i):
        r = self.transforms[i]
        return TransformList(r) if isinstance(i, slice) else r

    def __add__(self, other):
        return TransformList(self.transforms + (other.transforms if isinstance(other, TransformList) else [other]))


class Augmentation:
    def get_transform(self, image):
        raise NotImplementedError

    def __repr__(self):
        return type(self).__name__ + "(" + ", ".join(f"{k}={v}" for k, v in self.__dict__.items()) + ")"


class ResizeShortestEdge(Augmentation):
    def __init__(self, short_edge_length, max_size=2 ** 31 - 1, sample_style="range", interp=None):
        assert sample_style in ("range", "choice"), sample_style
        if isinstance(short_edge_length, int):
            short_edge_length = (short_edge_length, short_edge_length)
        if sample_style == "range":
            assert len(short_edge_length) == 2, "short_edge_length must be two values using 'range' sample style."
        self.short_edge_length, self.max_size, self.sample_style, self.interp = \
            tuple(short_edge_length), max_size, sample_style, interp

    @staticmethod
    def get_output_shape(oldh, oldw, short_edge_length, max_size):
        h, w = oldh, oldw
        scale = short_edge_length * 1.0 / min(h, w)
        if h < w:
            newh, neww = short_edge_length * 1.0, scale * w
        else:
            newh, neww = scale * h, short_edge_length * 1.0
        if max(newh, neww) > max_size:
            scale = max_size * 1.0 / max(newh, neww)
            newh, neww = newh * scale, neww * scale
        return int(newh + 0.5), int(neww + 0.5)

    def get_transform(self, image):
        h, w = image.shape[:2]
        if self.sample_style == "range":
            size = np.random.randint(self.short_edge_length[0], self.short_edge_length[1] + 1)
        else:
            size = np.random.choice(self.short_edge_length)
        if size == 0:
            return NoOpTransform()
        newh, neww = self.get_output_shape(h, w, size, self.max_size)
        return ResizeTransform(h, w, newh, neww, self.interp)


class RandomFlip(Augmentation):
    def __init__(self, prob=0.5, *, horizontal=True, vertical=False):
        if horizontal and vertical:
            raise ValueError("Cannot do both horiz and vert. Please use two Flip instead.")
        if not horizontal and not vertical:
            raise ValueError("At least one of horiz or vert has to be True!")
        self.prob, self.horizontal, self.vertical = prob, horizontal, vertical

    def get_transform(self, image):
        h, w = image.shape[:2]
        if np.random.uniform() < self.prob:
            return HFlipTransform(w) if self.horizontal else VFlipTransform(h)
        return NoOpTransform()


class AugInput:
    def __init__(self, image):
        self.image = image


class AugmentationList(Augmentation):
    """applies the augmentations in order to `aug_input.image` (in place) and returns the TransformList."""

    def __init__(self, augs):
        self.augs = list(augs)

    def __call__(self, aug_input):
        tfms = []
        for a in self.augs:
            t = a if isinstance(a, Transform) else a.get_transform(aug_input.image)
            aug_input.image = t.apply_image(aug_input.image)
            tfms.append(t)
        return TransformList(tfms)

    def __repr__(self):
        return repr(self.augs)


def build_augmentation(cfg, is_train):
    if is_train:
        min_size, max_size, style = cfg.INPUT.MIN_SIZE_TRAIN, cfg.INPUT.MAX_SIZE_TRAIN, cfg.INPUT.MIN_SIZE_TRAIN_SAMPLING
    else:
        min_size, max_size, style = cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST, "choice"
    if isinstance(min_size, int):
        min_size = (min_size,)
    augs = [ResizeShortestEdge(tuple(min_size), max_size, style)]
    if is_train and cfg.INPUT.RANDOM_FLIP != "none":
        augs.append(RandomFlip(horizontal=cfg.INPUT.RANDOM_FLIP == "horizontal",
                               vertical=cfg.INPUT.RANDOM_FLIP == "vertical"))
    return augs


# ------------------------------------------------------------------ image io
class SizeMismatchError(ValueError):
    pass


def read_image(file_name, format=None):
    """HWC uint8; "BGR" flips the channel order of the decoded RGB image.  `.npy` files hold the array as is."""
    if str(file_name).endswith(".npy"):
        return np.load(file_name)
    from PIL import Image, ImageOps
    with open(file_name, "rb") as f:
        image = Image.open(f)
        image = ImageOps.exif_transpose(image)
        conv = format
        if format == "BGR":
            conv = "RGB"
        if conv is not None:
            image = image.convert(conv)
        image = np.asarray(image)
    if format == "L":
        image = np.expand_dims(image, -1)
    elif format == "BGR":
        image = image[:, :, ::-1]
    return image


def check_image_size(dataset_dict, image):
    if "width" in dataset_dict or "height" in dataset_dict:
        image_wh = (image.shape[1], image.shape[0])
        expected_wh = (dataset_dict["width"], dataset_dict["height"])
        if image_wh != expected_wh:
            raise SizeMismatchError("Mismatched image shape{}, got {}, expect {}.".format(
                " for image " + dataset_dict["file_name"] if "file_name" in dataset_dict else "", image_wh, expected_wh))
    if "width" not in dataset_dict:
        dataset_dict["width"] = image.shape[1]
    if "height" not in dataset_dict:
        dataset_dict["height"] = image.shape[0]


def filter_empty_instances(instances, by_box=True, box_threshold=1e-5):
    if not instances.has("gt_boxes") or not by_box:
        return instances
    keep = instances.gt_boxes.nonempty(threshold=box_threshold)
    return instances[keep]


# ------------------------------------------------------------------ samplers
def _dist_rank_world(rank=None, world_size=None):
    if rank is not None and world_size is not None:
        return rank, world_size
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class TrainingSampler(tud.Sampler):
    """infinite stream of indices; every rank draws the same seeded permutations and keeps every world_size-th."""

    def __init__(self, size, shuffle=True, seed=0, rank=None, world_size=None):
        assert size > 0
        self._size, self._shuffle, self._seed = size, shuffle, int(seed)
        self._rank, self._world_size = _dist_rank_world(rank, world_size)

    def __iter__(self):
        yield from itertools.islice(self._infinite_indices(), self._rank, None, self._world_size)

    def _infinite_indices(self):
        g = torch.Generator()
        g.manual_seed(self._seed)
        while True:
            if self._shuffle:
                yield from torch.randperm(self._size, generator=g).tolist()
            else:
                yield from torch.arange(self._size).tolist()


class RepeatFactorTrainingSampler(tud.Sampler):
    """image i appears floor(r_i) times per epoch plus once more with probability frac(r_i)."""

    def __init__(self, repeat_factors, *, shuffle=True, seed=0, rank=None, world_size=None):
        self._shuffle, self._seed = shuffle, int(seed)
        self._rank, self._world_size = _dist_rank_world(rank, world_size)
        self._int_part = torch.trunc(repeat_factors)
        self._frac_part = repeat_factors - self._int_part

    def _get_epoch_indices(self, generator):
        rands = torch.rand(len(self._frac_part), generator=generator)
        rep = self._int_part + (rands < self._frac_part).float()
        indices = []
        for i, r in enumerate(rep):
            indices.extend([i] * int(r.item()))
        return torch.tensor(indices, dtype=torch.int64)

    def __iter__(self):
        yield from itertools.islice(self._infinite_indices(), self._rank, None, self._world_size)

    def _infinite_indices(self):
        g = torch.Generator()
        g.manual_seed(self._seed)
        while True:
            indices = self._get_epoch_indices(g)
            if self._shuffle:
                yield from indices[torch.randperm(len(indices), generator=g)].tolist()
            else:
                yield from indices.tolist()


class InferenceSampler(tud.Sampler):
    """contiguous shards, the first `size % world_size` ranks get one more."""

    def __init__(self, size, rank=None, world_size=None):
        assert size > 0
        self._size = size
        self._rank, self._world_size = _dist_rank_world(rank, world_size)
        self._local_indices = self._get_local_indices(size, self._world_size, self._rank)

    @staticmethod
    def _get_local_indices(total_size, world_size, rank):
        shard, left = total_size // world_size, total_size % world_size
        sizes = [shard + int(r < left) for r in range(world_size)]
        begin = sum(sizes[:rank])
        return range(begin, min(begin + sizes[rank], total_size))

    def __iter__(self):
        yield from self._local_indices

    def __len__(self):
        return len(self._local_indices)


# ------------------------------------------------------------------ datasets / loaders
class DatasetFromList(tud.Dataset):
    def __init__(self, lst, copy=True):
        self._lst, self._copy = lst, copy

    def __len__(self):
        return len(self._lst)

    def __getitem__(self, idx):
        if self._copy:
            import copy as _copy
            return _copy.deepcopy(self._lst[idx])
        return self._lst[idx]


class MapDataset(tud.Dataset):
    """applies `map_func`; an element mapped to None is replaced by another (seeded) element, as in detectron2."""

    def __init__(self, dataset, map_func):
        import random
        self._dataset, self._map_func = dataset, map_func
        self._rng = random.Random(42)
        self._fallback_candidates = set(range(len(dataset)))

    def __len__(self):
        return len(self._dataset)

    def __getitem__(self, idx):
        retry, cur = 0, int(idx)
        while True:
            data = self._map_func(self._dataset[cur])
            if data is not None:
                self._fallback_candidates.add(cur)
                return data
            retry += 1
            self._fallback_candidates.discard(cur)
            cur = self._rng.sample(sorted(self._fallback_candidates), k=1)[0]
            if retry >= 3:
                import logging
                logging.getLogger(__name__).warning("Failed to apply `_map_func` for idx: {}, retry count: {}".format(idx, retry))


class _SampledIterable(tud.IterableDataset):
    """map-style dataset + sampler -> stream; loader workers take interleaved elements of the sampler's stream."""

    def __init__(self, dataset, sampler):
        self.dataset, self.sampler = dataset, sampler

    def __iter__(self):
        info = tud.get_worker_info()
        it = iter(self.sampler)
        if info is not None and info.num_workers > 1:
            it = itertools.islice(it, info.id, None, info.num_workers)
        for idx in it:
            yield self.dataset[idx]


class AspectRatioGroupedDataset(tud.IterableDataset):
    """batches of `batch_size` elements that are all landscape (w > h) or all portrait."""

    def __init__(self, dataset, batch_size):
        self.dataset, self.batch_size = dataset, batch_size
        self._buckets = [[] for _ in range(2)]

    def __iter__(self):
        for d in self.dataset:
            bucket = self._buckets[0 if d["width"] > d["height"] else 1]
            bucket.append(d)
            if len(bucket) == self.batch_size:
                data = bucket[:]
                del bucket[:]
                yield data


def trivial_batch_collator(batch):
    return batch


def worker_init_reset_seed(worker_id):
    """detectron2.data.build.worker_init_reset_seed: loader workers are forked with identical numpy / random states, so
    every worker would draw the same augmentation sequence; reseed them from torch's per-worker seed"""
    import random
    seed = (torch.initial_seed() % (2 ** 31)) + worker_id
    np.random.seed(seed % (2 ** 31))
    random.seed(seed)


def _first(batch):
    return batch[0]


def build_batch_data_loader(dataset, sampler, total_batch_size, *, aspect_ratio_grouping=False, num_workers=0,
                            world_size=None):
    """per-rank loader yielding lists of `total_batch_size / world_size` mapped dicts (SOLVER.IMS_PER_BATCH is the
    GLOBAL batch)."""
    if world_size is None:
        world_size = _dist_rank_world()[1]
    assert total_batch_size > 0 and total_batch_size % world_size == 0, \
        "Total batch size ({}) must be divisible by the number of gpus ({}).".format(total_batch_size, world_size)
    batch_size = total_batch_size // world_size
    stream = _SampledIterable(dataset, sampler)
    if aspect_ratio_grouping:
        loader = tud.DataLoader(stream, num_workers=num_workers, batch_size=None, collate_fn=None,
                                worker_init_fn=worker_init_reset_seed)
        return AspectRatioGroupedDataset(loader, batch_size)
    return tud.DataLoader(stream, batch_size=batch_size, drop_last=True, num_workers=num_workers,
                          collate_fn=trivial_batch_collator, worker_init_fn=worker_init_reset_seed)
