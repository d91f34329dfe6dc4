"""Category sets of Omni3D and its source datasets (reference: cubercnn/data/builtin.py:3-45; the table is data).
Names are listed alphabetically; the sets are what the evaluator uses to pick the per-dataset categories."""

_CATEGORIES = (
    (('omni3d',),
     ('barrier', 'bathtub', 'bed', 'bicycle', 'bin', 'blinds', 'bookcase', 'books', 'bottle', 'box', 'bus',
     'cabinet', 'camera', 'car', 'cereal box', 'chair', 'clothes', 'counter', 'cup', 'curtain', 'cyclist',
     'desk', 'door', 'floor mat', 'lamp', 'laptop', 'machine', 'mirror', 'motorcycle', 'night stand', 'oven',
     'pedestrian', 'picture', 'pillow', 'refrigerator', 'shelves', 'shoes', 'sink', 'sofa', 'stationery',
     'stove', 'table', 'television', 'toilet', 'towel', 'traffic cone', 'trailer', 'truck', 'van', 'window')),
    (('omni3d_in',),
     ('bathtub', 'bed', 'bicycle', 'bin', 'blinds', 'bookcase', 'books', 'bottle', 'box', 'cabinet', 'chair',
     'clothes', 'counter', 'cup', 'curtain', 'desk', 'door', 'floor mat', 'lamp', 'laptop', 'machine', 'mirror',
     'night stand', 'oven', 'picture', 'pillow', 'refrigerator', 'shelves', 'shoes', 'sink', 'sofa',
     'stationery', 'stove', 'table', 'television', 'toilet', 'towel', 'window')),
    (('omni3d_out',),
     ('barrier', 'bicycle', 'bus', 'car', 'cyclist', 'motorcycle', 'pedestrian', 'traffic cone', 'trailer',
     'truck', 'van')),
    (('SUNRGBD_train', 'SUNRGBD_val', 'SUNRGBD_test', 'SUNRGBD_train_mini', 'SUNRGBD_val_mini', 'SUNRGBD_test_mini', 'SUNRGBD_test_mini2', 'SUNRGBD_test_mini500'),
     ('bathtub', 'bed', 'bicycle', 'bin', 'blinds', 'bookcase', 'books', 'bottle', 'box', 'cabinet', 'chair',
     'clothes', 'counter', 'cup', 'curtain', 'desk', 'door', 'floor mat', 'lamp', 'laptop', 'machine', 'mirror',
     'night stand', 'oven', 'picture', 'pillow', 'refrigerator', 'shelves', 'shoes', 'sink', 'sofa',
     'stationery', 'stove', 'table', 'television', 'toilet', 'towel', 'window')),
    (('Hypersim_train', 'Hypersim_val'),
     ('bathtub', 'bed', 'blinds', 'bookcase', 'books', 'box', 'cabinet', 'chair', 'clothes', 'counter', 'curtain',
     'desk', 'door', 'floor mat', 'lamp', 'mirror', 'night stand', 'picture', 'pillow', 'refrigerator',
     'shelves', 'sink', 'sofa', 'stationery', 'table', 'television', 'toilet', 'towel', 'window')),
    (('Hypersim_test',),
     ('bathtub', 'bed', 'blinds', 'bookcase', 'books', 'box', 'cabinet', 'chair', 'clothes', 'counter', 'curtain',
     'desk', 'door', 'floor mat', 'lamp', 'mirror', 'night stand', 'picture', 'pillow', 'refrigerator',
     'shelves', 'sink', 'sofa', 'stationery', 'table', 'television', 'towel', 'window')),
    (('ARKitScenes_train', 'ARKitScenes_val', 'ARKitScenes_test'),
     ('bathtub', 'bed', 'cabinet', 'chair', 'machine', 'oven', 'refrigerator', 'shelves', 'sink', 'sofa', 'stove',
     'table', 'television', 'toilet')),
    (('Objectron_train', 'Objectron_val', 'Objectron_test'),
     ('bicycle', 'books', 'bottle', 'camera', 'cereal box', 'chair', 'cup', 'laptop', 'shoes')),
    (('KITTI_train', 'KITTI_val', 'KITTI_test'),
     ('car', 'cyclist', 'pedestrian', 'truck', 'van')),
    (('nuScenes_train', 'nuScenes_val', 'nuScenes_test'),
     ('barrier', 'bicycle', 'bus', 'car', 'motorcycle', 'pedestrian', 'traffic cone', 'trailer', 'truck')),
)

_BY_NAME = {name: frozenset(cats) for names, cats in _CATEGORIES for name in names}


def get_omni3d_categories(dataset="omni3d"):
    """set of category names annotated in `dataset` ("omni3d", "omni3d_in", "omni3d_out" or a source split such as
    "KITTI_test"); unknown names raise ValueError like the reference."""
    if dataset not in _BY_NAME:
        raise ValueError("%s dataset is not registered." % (dataset))
    return set(_BY_NAME[dataset])
