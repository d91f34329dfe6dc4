"""Loaders of the Omni3D data path (reference: cubercnn/data/build.py) and the device staging that feeds the train
step on an MI355X.

One process per GPU: every rank builds the same dataset list and the same seeded sampler and keeps its own
interleaved share of the index stream (`SOLVER.IMS_PER_BATCH` is the global batch; a rank's batch is
IMS_PER_BATCH / world_size images).  `DevicePrefetcher` copies the next batch's uint8 images and GT tensors from pinned
host memory on a side HIP stream while the current step runs, so the H2D copy (4 x 768 KB at 512x512) is off the
step's critical path.
"""
import itertools
import logging
import math
from collections import defaultdict

import numpy as np
import ctypes

import torch
import torch.utils.data as tud

from ...d2lite import data as D
from ...d2lite.data import DatasetCatalog


def filter_images_with_only_crowd_annotations(dataset_dicts):
    """build.py:25-53"""
    before = len(dataset_dicts)
    dataset_dicts = [d for d in dataset_dicts if any(a.get("iscrowd", 0) == 0 for a in d["annotations"])]
    logging.getLogger(__name__).info(
        "Removed {} images marked with crowd. {} images left.".format(before - len(dataset_dicts), len(dataset_dicts)))
    return dataset_dicts


def get_detection_dataset_dicts(names, filter_empty=True, **kwargs):
    """build.py:55-74: concatenation of the registered datasets."""
    if isinstance(names, str):
        names = [names]
    assert len(names), names
    per_name = [DatasetCatalog.get(n) for n in names]
    for n, dicts in zip(names, per_name):
        assert len(dicts), "Dataset '{}' is empty!".format(n)
    dataset_dicts = list(itertools.chain.from_iterable(per_name))
    if filter_empty and "annotations" in dataset_dicts[0]:
        dataset_dicts = filter_images_with_only_crowd_annotations(dataset_dicts)
    assert len(dataset_dicts), "No valid data found in {}.".format(",".join(names))
    return dataset_dicts


def repeat_factors_from_category_frequency(dataset_dicts, repeat_thresh):
    """build.py:154-202 (LVIS repeat-factor sampling; ignored annotations, class < 0, do not count):
    f(c) = fraction of images containing c, r(c) = max(1, sqrt(t / f(c))), r(image) = max over its classes."""
    n_images = len(dataset_dicts)
    per_image = [{a["category_id"] for a in d["annotations"] if a["category_id"] >= 0} for d in dataset_dicts]
    count = defaultdict(int)
    for cats in per_image:
        for c in cats:
            count[c] += 1
    rep = {c: max(1.0, math.sqrt(repeat_thresh / (v / n_images))) for c, v in count.items()}
    return torch.tensor([max((rep[c] for c in cats), default=1.0) for cats in per_image], dtype=torch.float32)


def dataset_balance_weights(dataset, dataset_id_to_src):
    """build.py:96-121: per-image weight 1 - share(source), normalised so the most frequent source has weight 1."""
    src_to_int = {v: i for i, v in enumerate(set(dataset_id_to_src.values()))}
    ids = [src_to_int[dataset_id_to_src[img['dataset_id']]] for img in dataset]
    uniq = np.unique(ids)
    if len(uniq) == 1:
        return torch.ones(len(ids)).float()
    counts = np.bincount(ids)
    counts = [counts[i] for i in uniq]
    weights = [1 - c / np.sum(counts) for c in counts]
    weights = [w / np.min(weights) for w in weights]
    out = torch.zeros(len(ids)).float()
    ids_t = torch.tensor(ids, dtype=torch.int64)
    for i, w in zip(uniq, weights):
        out[ids_t == int(i)] = float(w)
    return out


def _train_sampler(cfg, dataset, dataset_id_to_src, seed, rank, world_size):
    """build.py:88-142"""
    name, balance = cfg.DATALOADER.SAMPLER_TRAIN, cfg.DATALOADER.BALANCE_DATASETS
    logging.getLogger(__name__).info("Using training sampler {}".format(name))
    kw = dict(seed=seed, rank=rank, world_size=world_size)
    if balance:
        assert dataset_id_to_src is not None, 'Need dataset sources.'
        w = dataset_balance_weights(dataset, dataset_id_to_src)
    if name == "TrainingSampler":
        return D.RepeatFactorTrainingSampler(w, **kw) if balance else D.TrainingSampler(len(dataset), **kw)
    if name == "RepeatFactorTrainingSampler":
        rf = repeat_factors_from_category_frequency(dataset, cfg.DATALOADER.REPEAT_THRESHOLD)
        if balance:
            rf = rf * w
            rf = rf / rf.min().item()
        return D.RepeatFactorTrainingSampler(rf, **kw)
    raise ValueError("Unknown training sampler: {}".format(name))


def build_detection_train_loader(cfg=None, mapper=None, *, dataset=None, sampler=None, dataset_id_to_src=None,
                                 total_batch_size=None, aspect_ratio_grouping=None, num_workers=None, seed=None,
                                 rank=None, world_size=None):
    """build.py:77-152,204-221.  Infinite iterator over lists of mapped dicts (this rank's share of the global batch)."""
    rank, world_size = D._dist_rank_world(rank, world_size)
    if dataset is None:
        dataset = get_detection_dataset_dicts(cfg.DATASETS.TRAIN, filter_empty=cfg.DATALOADER.FILTER_EMPTY_ANNOTATIONS)
    if mapper is None:
        from .dataset_mapper import DatasetMapper3D
        mapper = DatasetMapper3D(cfg, True)
    if seed is None:
        seed = int(cfg.SEED) if cfg is not None and int(cfg.get("SEED", -1)) >= 0 else 0
    if sampler is None:
        sampler = _train_sampler(cfg, dataset, dataset_id_to_src, seed, rank, world_size) if cfg is not None \
            else D.TrainingSampler(len(dataset), seed=seed, rank=rank, world_size=world_size)
    if total_batch_size is None:
        total_batch_size = cfg.SOLVER.IMS_PER_BATCH
    if aspect_ratio_grouping is None:
        aspect_ratio_grouping = cfg.DATALOADER.ASPECT_RATIO_GROUPING if cfg is not None else True
    if num_workers is None:
        num_workers = cfg.DATALOADER.NUM_WORKERS if cfg is not None else 0
    if isinstance(dataset, list):
        dataset = D.DatasetFromList(dataset, copy=False)
    if mapper is not None:
        dataset = D.MapDataset(dataset, mapper)
    assert isinstance(sampler, tud.Sampler)
    return D.build_batch_data_loader(dataset, sampler, total_batch_size, aspect_ratio_grouping=aspect_ratio_grouping,
                                     num_workers=num_workers, world_size=world_size)


def build_detection_test_loader(cfg=None, dataset_name=None, *, dataset=None, mapper=None, batch_size=1, sampler=None,
                                num_workers=None, filter_empty=False, rank=None, world_size=None):
    """build.py:223-260: this rank's contiguous shard of the dataset, in order, `batch_size` images per step."""
    if dataset is None:
        dataset = get_detection_dataset_dicts(dataset_name, filter_empty=filter_empty)
    if mapper is None:
        from .dataset_mapper import DatasetMapper3D
        mapper = DatasetMapper3D(cfg, False)
    if num_workers is None:
        num_workers = cfg.DATALOADER.NUM_WORKERS if cfg is not None else 0
    if isinstance(dataset, list):
        dataset = D.DatasetFromList(dataset, copy=False)
    if mapper is not None:
        dataset = D.MapDataset(dataset, mapper)
    if sampler is None:
        sampler = D.InferenceSampler(len(dataset), rank=rank, world_size=world_size)
    batch_sampler = tud.BatchSampler(sampler, batch_size=batch_size, drop_last=False)
    return tud.DataLoader(dataset, num_workers=num_workers, batch_sampler=batch_sampler,
                          collate_fn=D.trivial_batch_collator, worker_init_fn=D.worker_init_reset_seed)


def dataset_id_maps(datasets, num_classes, id_map):
    """tools/train_net.py:418-446: per source dataset, the contiguous class ids it does NOT annotate (the extra index
    `num_classes` = background is always 'unknown' unless annotated), and dataset id -> source name."""
    infos = datasets.dataset['info']
    if type(infos) == dict:
        infos = [infos]
    possible = set(range(num_classes + 1))
    unknown_cats, id_to_src = {}, {}
    for info in infos:
        did = info['id']
        id_to_src.setdefault(did, info['source'])
        known = {id_map[i] for i in info['known_category_ids'] if i in id_map}
        unknown_cats[did] = possible - known
    return unknown_cats, id_to_src


class DevicePrefetcher:
    """Wraps a loader of list[dict] batches: the host tensors of batch i+1 ('image', 'depth_map', 'ground_map' and the fields
    of 'instances') are packed into ONE pinned staging buffer and copied to `device` with ONE asynchronous H2D transfer on
    a side stream while batch i is in use; the batch's tensors are views into that device buffer.

    The staging buffers are a ring of three persistent pinned allocations (grown on demand): `Tensor.pin_memory()` per
    tensor costs a hipHostMalloc + hipHostFree each -- ~20 per batch, every one a device-wide synchronisation -- which made
    the loop twice as slow as the step it feeds (35 ms vs 17 ms per 4-image step).

    The consumer's stream waits on the copy's event, and the device buffer is recorded on it, so the caching allocator does
    not recycle it while the step still reads it."""

    TENSOR_KEYS = ("image", "depth_map", "ground_map")
    ALIGN = 256

    def __init__(self, loader, device):
        self.device = torch.device(device)
        assert self.device.type == "cuda", "DevicePrefetcher stages onto a GPU"
        self._it = iter(loader)
        self._stream = torch.cuda.Stream(device=self.device)
        self._ring = [[None, None] for _ in range(3)]          # [pinned uint8 buffer, event of the last copy out of it]
        self._turn = 0
        self._next = None
        self._preload()

    def _staging(self, nbytes):
        slot = self._ring[self._turn]
        self._turn = (self._turn + 1) % len(self._ring)
        if slot[1] is not None:
            slot[1].synchronize()                               # the copy that last read this buffer (3 batches ago) is done
        if slot[0] is None or slot[0].numel() < nbytes:
            slot[0] = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8).pin_memory()
        return slot

    def _preload(self):
        try:
            batch = next(self._it)
        except StopIteration:
            self._next = None
            return
        out, jobs, total = [], [], 0

        def want(t, assign):
            """host tensor -> a slot of the packed buffer; `assign(device_view)` puts the result where the tensor was"""
            nonlocal total
            if not isinstance(t, torch.Tensor) or t.is_cuda:
                assign(t)
                return
            jobs.append((t, total, assign))
            total += (t.numel() * t.element_size() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        for d in batch:
            d = dict(d)
            for k in self.TENSOR_KEYS:
                if d.get(k) is not None:
                    want(d[k], lambda v, d=d, k=k: d.__setitem__(k, v))
            inst = d.get("instances")
            if inst is not None:
                staged = type(inst)(inst.image_size)
                for name, v in inst.get_fields().items():
                    if isinstance(v, torch.Tensor):
                        want(v, lambda x, s=staged, n=name: s.set(n, x))
                    elif hasattr(v, "tensor"):
                        want(v.tensor, lambda x, s=staged, n=name, T=type(v): s.set(n, T(x)))
                    else:
                        staged.set(name, v)
                d["instances"] = staged
            out.append(d)
        dev_buf = None
        if jobs:
            slot = self._staging(total)
            pin = slot[0]
            base = pin.data_ptr()
            for t, off, _ in jobs:
                n = t.numel() * t.element_size()
                if n:
                    # plain memcpy on the calling thread: a torch copy_ of an image-sized CPU tensor opens an OpenMP region,
                    # and on a box whose CPU quota is far below its visible core count the spinning workers stall the
                    # launching thread for tens of ms every few batches (DESIGN: findings worth keeping)
                    tc = t if t.is_contiguous() else t.contiguous()
                    ctypes.memmove(base + off, tc.data_ptr(), n)
            with torch.cuda.stream(self._stream):
                dev_buf = torch.empty(total, dtype=torch.uint8, device=self.device)
                dev_buf.copy_(pin[:total], non_blocking=True)
                slot[1] = torch.cuda.Event()
                slot[1].record(self._stream)
            for t, off, assign in jobs:
                n = t.numel() * t.element_size()
                assign(dev_buf[off:off + n].view(t.dtype).view(t.shape))
        ev = torch.cuda.Event()
        ev.record(self._stream)
        self._next = (out, ev, dev_buf)

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        batch, ev, dev_buf = self._next
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        if dev_buf is not None:
            dev_buf.record_stream(cur)          # every tensor of the batch is a view of this one allocation
        self._preload()
        return batch
