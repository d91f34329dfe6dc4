"""Omni3D annotation files -> dataset dicts (reference: cubercnn/data/datasets.py).

The wire format is the Omni3D json of DATA.md:140-200: {"info", "images": [{id, dataset_id, width, height, file_path,
K}], "categories": [{id, name}], "annotations": [{id, image_id, category_id, category_name, valid3D, bbox2D_tight,
bbox2D_proj, bbox2D_trunc, bbox3D_cam (8x3), center_cam, dimensions (w,h,l), R_cam, behind_camera, visibility,
truncation, segmentation_pts, lidar_pts, depth_error}]}.

pycocotools is not installed, so the COCO index the reference inherits (`class Omni3D(COCO)`, datasets.py:140) is
restated here as `_CocoIndex` (createIndex / getAnnIds / loadAnns / getCatIds / loadCats / loadImgs)
[third-party, parity unpinned]; filtering and record building follow the reference and are pinned by
tests/golden/data_path.json (made by running the reference's functions, tests/golden/make_golden_data.py).
"""
import json
import logging
import os
from collections import defaultdict

import numpy as np

from ... import d2lite as _d2
from ...d2lite.data import BoxMode, DatasetCatalog, MetadataCatalog
from ..util import util

VERSION = '0.1'
logger = logging.getLogger(__name__)

_STATS_DEFAULT = os.path.join('datasets', 'Omni3D', 'stats.json')


def get_version():
    return VERSION


def get_global_dataset_stats(path_to_stats=None, reset=False):
    """datasets.py:26-43"""
    path_to_stats = path_to_stats or _STATS_DEFAULT
    if os.path.exists(path_to_stats) and not reset:
        return util.load_json(path_to_stats)
    return {'n_datasets': 0, 'n_ims': 0, 'n_anns': 0, 'categories': []}


def save_global_dataset_stats(stats, path_to_stats=None):
    util.save_json(path_to_stats or _STATS_DEFAULT, stats)


def get_filter_settings_from_cfg(cfg=None):
    """datasets.py:54-80.  `max_height_thres` is not a config key in the reference either."""
    fs = {
        'category_names': [], 'ignore_names': [],
        'truncation_thres': 0.99, 'visibility_thres': 0.01,
        'min_height_thres': 0.00, 'max_height_thres': 1.50,
        'modal_2D_boxes': False, 'trunc_2D_boxes': False, 'max_depth': 1e8,
    }
    if cfg is not None:
        D = cfg.DATASETS
        fs.update(category_names=D.CATEGORY_NAMES, ignore_names=D.IGNORE_NAMES, truncation_thres=D.TRUNCATION_THRES,
                  visibility_thres=D.VISIBILITY_THRES, min_height_thres=D.MIN_HEIGHT_THRES,
                  modal_2D_boxes=D.MODAL_2D_BOXES, trunc_2D_boxes=D.TRUNC_2D_BOXES, max_depth=D.MAX_DEPTH)
    return fs


def _xywh(xyxy):
    return BoxMode.convert(xyxy, BoxMode.XYXY_ABS, BoxMode.XYWH_ABS)


def _has_trunc_box(anno):
    return 'bbox2D_trunc' in anno and not all(v == -1 for v in anno['bbox2D_trunc'])


def is_ignore(anno, filter_settings, image_height):
    """datasets.py:83-123: an annotation is ignored (kept as a don't-care region, class -1) when its 3D box is not
    usable or the object is too small / large / truncated / occluded."""
    if anno['behind_camera'] or not bool(anno['valid3D']):
        return True
    dims = anno['dimensions']
    ignore = dims[0] <= 0.01 or dims[1] <= 0.01 or dims[2] <= 0.01
    ignore |= anno['center_cam'][2] > filter_settings['max_depth']
    ignore |= anno['lidar_pts'] == 0
    ignore |= anno['segmentation_pts'] == 0
    ignore |= anno['depth_error'] > 0.5

    # the 2D box the height test runs on: tight (modal) -> truncated projection -> projection -> given bbox
    if filter_settings['modal_2D_boxes'] and 'bbox2D_tight' in anno and anno['bbox2D_tight'][0] != -1:
        box = _xywh(anno['bbox2D_tight'])
    elif filter_settings['trunc_2D_boxes'] and _has_trunc_box(anno):
        box = _xywh(anno['bbox2D_trunc'])
    elif 'bbox2D_proj' in anno:
        box = _xywh(anno['bbox2D_proj'])
    else:
        box = anno['bbox']
    ignore |= box[3] <= filter_settings['min_height_thres'] * image_height
    ignore |= box[3] >= filter_settings['max_height_thres'] * image_height
    ignore |= (anno['truncation'] >= 0 and anno['truncation'] >= filter_settings['truncation_thres'])
    ignore |= (anno['visibility'] >= 0 and anno['visibility'] <= filter_settings['visibility_thres'])
    if 'ignore_names' in filter_settings:
        ignore |= anno['category_name'] in filter_settings['ignore_names']
    return bool(ignore)


class _CocoIndex:
    """the slice of pycocotools.coco.COCO the data path uses."""

    def __init__(self, annotation_file=None):
        self.dataset, self.anns, self.cats, self.imgs = {}, {}, {}, {}
        self.imgToAnns, self.catToImgs = defaultdict(list), defaultdict(list)
        if annotation_file is not None:
            with open(annotation_file, 'r') as f:
                dataset = json.load(f)
            assert type(dataset) == dict, 'annotation file format {} not supported'.format(type(dataset))
            self.dataset = dataset
            self.createIndex()

    def createIndex(self):
        anns, cats, imgs = {}, {}, {}
        imgToAnns, catToImgs = defaultdict(list), defaultdict(list)
        for ann in self.dataset.get('annotations', ()):
            imgToAnns[ann['image_id']].append(ann)
            anns[ann['id']] = ann
        for img in self.dataset.get('images', ()):
            imgs[img['id']] = img
        for cat in self.dataset.get('categories', ()):
            cats[cat['id']] = cat
        if 'categories' in self.dataset:
            for ann in self.dataset.get('annotations', ()):
                catToImgs[ann['category_id']].append(ann['image_id'])
        self.anns, self.imgToAnns, self.catToImgs, self.imgs, self.cats = anns, imgToAnns, catToImgs, imgs, cats

    @staticmethod
    def _aslist(x):
        return x if isinstance(x, (list, tuple, set)) else [x]

    def getAnnIds(self, imgIds=(), catIds=()):
        imgIds, catIds = self._aslist(imgIds), self._aslist(catIds)
        if len(imgIds):
            anns = [a for i in imgIds if i in self.imgToAnns for a in self.imgToAnns[i]]
        else:
            anns = self.dataset['annotations']
        if len(catIds):
            anns = [a for a in anns if a['category_id'] in catIds]
        return [a['id'] for a in anns]

    def getCatIds(self, catNms=()):
        catNms = self._aslist(catNms)
        cats = self.dataset['categories']
        if len(catNms):
            cats = [c for c in cats if c['name'] in catNms]
        return [c['id'] for c in cats]

    def getImgIds(self):
        return list(self.imgs.keys())

    def loadAnns(self, ids=()):
        return [self.anns[i] for i in ids] if isinstance(ids, (list, tuple)) else [self.anns[ids]]

    def loadCats(self, ids=()):
        return [self.cats[i] for i in ids] if isinstance(ids, (list, tuple)) else [self.cats[ids]]

    def loadImgs(self, ids=()):
        return [self.imgs[i] for i in ids] if isinstance(ids, (list, tuple)) else [self.imgs[ids]]


COCO = _CocoIndex


class Omni3D(_CocoIndex):
    """datasets.py:140-293: one or several annotation files merged into a COCO-like index, annotations filtered by
    `filter_settings` and given the evaluation fields (area, ignore*, bbox XYWH, bbox3D, depth).  Used by
    `compute_priors` and by the evaluator; independent of the dataset catalog."""

    def __init__(self, annotation_files, filter_settings=None, no_ground_csv=os.path.join('datasets', 'no_ground_idx.csv')):
        super().__init__()
        self.idx_without_ground = set()
        if no_ground_csv and os.path.exists(no_ground_csv):     # the reference requires this file (datasets.py:151)
            import pandas as pd
            self.idx_without_ground = set(pd.read_csv(no_ground_csv)['img_id'].values)
        if isinstance(annotation_files, str):
            annotation_files = [annotation_files]

        master = {}                                     # category id -> first definition seen
        for annotation_file in annotation_files:
            _, name, _ = util.file_parts(annotation_file)
            logger.info('loading {} annotations into memory...'.format(name))
            with open(annotation_file, 'r') as f:
                dataset = json.load(f)
            assert type(dataset) == dict, 'annotation file format {} not supported'.format(type(dataset))
            if type(dataset['info']) == list:
                dataset['info'] = dataset['info'][0]
            dataset['info']['known_category_ids'] = [cat['id'] for cat in dataset['categories']]
            if not self.dataset:
                self.dataset = dataset
            else:
                if type(self.dataset['info']) == dict:
                    self.dataset['info'] = [self.dataset['info']]
                self.dataset['info'] += [dataset['info']]
                self.dataset['annotations'] += dataset['annotations']
                self.dataset['images'] += dataset['images']
            for cat in dataset['categories']:
                master.setdefault(cat['id'], cat)
        cats_sorted = [master[i] for i in sorted(master)]

        if filter_settings is None:
            self.dataset['categories'] = cats_sorted
        else:
            trainable = set(filter_settings['ignore_names']) | set(filter_settings['category_names'])
            if len(filter_settings['category_names']) > 0:
                self.dataset['categories'] = [c for c in cats_sorted if c['name'] in filter_settings['category_names']]
            else:                                       # no names given: every category present is used
                self.dataset['categories'] = cats_sorted
                filter_settings['category_names'] = [c['name'] for c in cats_sorted]
                trainable |= set(filter_settings['category_names'])
            heights = {im['id']: im['height'] for im in self.dataset['images']}
            kept = []
            for anno in self.dataset['annotations']:
                ignore = is_ignore(anno, filter_settings, heights[anno['image_id']])
                # note the precedence differs from is_ignore's: truncated -> projected -> tight (datasets.py:236-246)
                if filter_settings['trunc_2D_boxes'] and _has_trunc_box(anno):
                    box = _xywh(anno['bbox2D_trunc'])
                elif anno['bbox2D_proj'][0] != -1:
                    box = _xywh(anno['bbox2D_proj'])
                elif anno['bbox2D_tight'][0] != -1:
                    box = _xywh(anno['bbox2D_tight'])
                else:
                    continue
                anno['area'] = box[2] * box[3]
                anno['iscrowd'] = False
                anno['ignore'] = anno['ignore2D'] = anno['ignore3D'] = ignore
                if filter_settings['modal_2D_boxes'] and anno['bbox2D_tight'][0] != -1:
                    anno['bbox'] = _xywh(anno['bbox2D_tight'])
                else:
                    anno['bbox'] = box
                anno['bbox3D'] = anno['bbox3D_cam']
                anno['depth'] = anno['center_cam'][2]
                if anno['category_name'] in trainable and not ignore:
                    kept.append(anno)
            self.dataset['annotations'] = kept
        self.createIndex()

    def info(self):
        infos = self.dataset['info']
        if type(infos) == dict:
            infos = [infos]
        for i, info in enumerate(infos):
            print('Dataset {}/{}'.format(i + 1, infos))
            for key, value in info.items():
                print('{}: {}'.format(key, value))


def register_and_store_model_metadata(datasets, output_dir, filter_settings=None, path_to_stats=None):
    """datasets.py:307-337: the model's class list and dataset-id -> contiguous-id map, written once to
    `<output_dir>/category_meta.json` and re-read on later runs (so a checkpoint keeps its class order)."""
    output_file = os.path.join(output_dir, 'category_meta.json')
    if os.path.exists(output_file):
        metadata = util.load_json(output_file)
        thing_classes = metadata['thing_classes']
        id_map = {int(a): b for a, b in metadata['thing_dataset_id_to_contiguous_id'].items()}   # json keys are strings
    else:
        stats = util.load_json(path_to_stats or _STATS_DEFAULT)
        names = list(filter_settings['category_names'])
        ids = [stats['categories'][stats['category_names'].index(n)]['id'] for n in names]
        order = np.argsort(ids)
        ids = [ids[i] for i in order]
        thing_classes = [names[i] for i in order]
        id_map = {cid: i for i, cid in enumerate(ids)}
        util.save_json(output_file, {'thing_classes': thing_classes, 'thing_dataset_id_to_contiguous_id': id_map})
    MetadataCatalog.get('omni3d_model').thing_classes = thing_classes
    MetadataCatalog.get('omni3d_model').thing_dataset_id_to_contiguous_id = id_map


def _indexed_maps(directory):
    """ids of the `<id>.npz` files in a directory (depth / ground maps)."""
    ids = set()
    if directory and os.path.isdir(directory):
        for name in os.listdir(directory):
            try:
                ids.add(int(name.split('.')[0]))
            except ValueError:
                pass
    return ids


def load_omni3d_json(json_file, image_root, dataset_name, filter_settings, filter_empty=True,
                     depth_dir=os.path.join('datasets', 'depth_maps'), ground_dir=os.path.join('datasets', 'ground_maps')):
    """datasets.py:339-479: list of records {file_name, dataset_id, height, width, K, image_id, [p2],
    [depth_image_path], [ground_image_path], annotations: [{bbox (XYWH), bbox_mode, bbox3D_cam, center_cam, dimensions,
    pose (= R_cam), category_id (contiguous, -1 = ignore), iscrowd, ignore, ...}]}."""
    coco_api = _CocoIndex(json_file)
    ground_idx, depth_idx = _indexed_maps(ground_dir), _indexed_maps(depth_dir)

    meta_model = MetadataCatalog.get('omni3d_model')
    meta = MetadataCatalog.get(dataset_name)
    cat_ids = sorted(coco_api.getCatIds(filter_settings['category_names']))
    cats = coco_api.loadCats(cat_ids)
    meta.thing_classes = [c["name"] for c in sorted(cats, key=lambda x: x["id"])]
    id_map = meta_model.thing_dataset_id_to_contiguous_id        # the id mapping is the MODEL's, not the file's
    meta.thing_dataset_id_to_contiguous_id = id_map

    img_ids = sorted(coco_api.imgs.keys())
    imgs = coco_api.loadImgs(img_ids)
    anns = [coco_api.imgToAnns[i] for i in img_ids]
    n_valid, n_all = sum(len(x) for x in anns), len(coco_api.anns)
    if n_valid < n_all:
        logger.info(f"{json_file} contains {n_all} annotations, but only {n_valid} of them match to images in the file.")
    logger.info("Loaded {} images in Omni3D format from {}".format(len(imgs), json_file))

    passthrough = ("bbox", "bbox3D_cam", "bbox2D_proj", "bbox2D_trunc", "bbox2D_tight", "center_cam", "dimensions",
                   "pose", "R_cam", "category_id")
    records, n_dropped = [], 0
    for img, img_anns in zip(imgs, anns):
        rec = {"file_name": os.path.join(image_root, img["file_path"]), "dataset_id": img["dataset_id"],
               "height": img["height"], "width": img["width"], "K": img["K"]}
        if 'p2' in img:                                  # KITTI only
            rec['p2'] = img['p2']
        image_id = rec["image_id"] = img["id"]
        if image_id in depth_idx:
            rec["depth_image_path"] = os.path.join(depth_dir, f'{image_id}.npz')
        if image_id in ground_idx:
            rec["ground_image_path"] = os.path.join(ground_dir, f'{image_id}.npz')
        objs, any_valid = [], False
        for anno in img_anns:
            assert anno["image_id"] == image_id
            obj = {k: anno[k] for k in passthrough if k in anno}
            obj["bbox_mode"] = BoxMode.XYWH_ABS
            cid = obj["category_id"]
            if cid not in id_map and anno['category_name'] not in filter_settings['ignore_names']:
                continue
            ignore = is_ignore(anno, filter_settings, img["height"])
            obj['iscrowd'] = False
            obj['ignore'] = ignore
            if filter_settings['modal_2D_boxes'] and 'bbox2D_tight' in anno and anno['bbox2D_tight'][0] != -1:
                obj['bbox'] = _xywh(anno['bbox2D_tight'])
            elif filter_settings['trunc_2D_boxes'] and _has_trunc_box(anno):
                obj['bbox'] = _xywh(anno['bbox2D_trunc'])
            elif 'bbox2D_proj' in anno:
                obj['bbox'] = _xywh(anno['bbox2D_proj'])
            else:
                continue
            obj['pose'] = anno['R_cam']
            obj["category_id"] = -1 if ignore else id_map[cid]       # ignored regions carry class -1
            objs.append(obj)
            any_valid |= not ignore
        if any_valid or not filter_empty:
            rec["annotations"] = objs
            records.append(rec)
        else:
            n_dropped += 1
    logger.info("Filtered out {}/{} images without valid annotations".format(n_dropped, len(imgs)))
    return records


def simple_register(dataset_name, filter_settings, filter_empty=True, datasets_root_path=None, image_root='datasets',
                    **load_kwargs):
    """datasets.py:126-138"""
    if datasets_root_path is None:
        datasets_root_path = os.path.join('datasets', 'Omni3D')
    path_to_json = os.path.join(datasets_root_path, dataset_name + '.json')
    DatasetCatalog.register(dataset_name, lambda: load_omni3d_json(
        path_to_json, image_root, dataset_name, filter_settings, filter_empty=filter_empty, **load_kwargs))
    MetadataCatalog.get(dataset_name).set(json_file=path_to_json, image_root=image_root, evaluator_type="coco")
