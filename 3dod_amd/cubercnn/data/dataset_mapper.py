"""Per-image mapper of the Omni3D data path (reference: cubercnn/data/dataset_mapper.py).

dataset dict -> {image (3,h,w) uint8 BGR tensor, height, width, K, [depth_map], [ground_map], instances} where
`instances` carries gt_classes, gt_boxes (XYXY), gt_boxes3D = [cx2d, cy2d, z, w, h, l, X, Y, Z]
(dataset_mapper.py:258), gt_poses (3x3), gt_keypoints (8 projected corners + visibility) and
gt_unknown_category_mask.  The image is left uint8: normalisation happens on the GPU (`cr_preprocess`), so the host
moves 1 byte per pixel-channel over PCIe instead of 4.
"""
import copy
import logging

import numpy as np
import torch

from ...d2lite import Boxes, Instances
from ...d2lite import data as D
from ...d2lite.data import BoxMode, Keypoints

# mirror of a rotation under a horizontal image flip (dataset_mapper.py:180-189): R' = M1 R M2
_M1 = np.diag([1.0, -1.0, -1.0])
_M2 = np.diag([-1.0, -1.0, 1.0])


class DatasetMapper3D:
    """dataset_mapper.py:24-172.  `mode` is kept for signature parity; crop / mask / keypoint-dataset options of the
    detectron2 base mapper are not part of any Cube R-CNN config and are rejected."""

    def __init__(self, cfg=None, is_train=True, *, augmentations=None, image_format="BGR", mode=None, only_2d=False):
        if cfg is not None:
            kw = self.from_config(cfg, is_train, mode or 'get_depth_maps')
            augmentations, image_format, only_2d, mode = kw["augmentations"], kw["image_format"], kw["only_2d"], kw["mode"]
        self.is_train = is_train
        self.augmentations = D.AugmentationList(augmentations or [])
        self.image_format = image_format
        self.only_2d = only_2d
        self.mode = mode
        self.dataset_id_to_unknown_cats = None          # set by the trainer (tools/train_net.py:147)
        logging.getLogger(__name__).info("[DatasetMapper] Augmentations used in {}: {}".format(
            "training" if is_train else "inference", augmentations))

    @classmethod
    def from_config(cls, cfg, is_train=True, mode='get_depth_maps'):
        crop = cfg.INPUT.get("CROP", None) if hasattr(cfg.INPUT, "get") else None
        if crop is not None and crop.get("ENABLED", False) and is_train:
            raise NotImplementedError("INPUT.CROP is not used by any Cube R-CNN config and is not built")
        return {"is_train": is_train, "mode": mode, "augmentations": D.build_augmentation(cfg, is_train),
                "image_format": cfg.INPUT.FORMAT, "only_2d": cfg.MODEL.ROI_CUBE_HEAD.LOSS_W_3D == 0.0}

    @staticmethod
    def _load_map(path, key, hw):
        """`<id>.npz` map resized (nearest) to the augmented image's size.  As in the reference the flip is NOT
        applied to the maps (dataset_mapper.py:129-149 leaves `transforms_dp` commented out)."""
        from PIL import Image
        with np.load(path) as f:
            m = Image.fromarray(f[key])
        m = np.array(m.resize((hw[1], hw[0]), Image.NEAREST))
        return torch.as_tensor(np.ascontiguousarray(m))

    def __call__(self, dataset_dict):
        dataset_dict = copy.deepcopy(dataset_dict)
        image = D.read_image(dataset_dict["file_name"], format=self.image_format)
        D.check_image_size(dataset_dict, image)
        aug_input = D.AugInput(image)
        transforms = self.augmentations(aug_input)
        image = aug_input.image
        image_shape = image.shape[:2]

        if not self.only_2d:
            dataset_dict["depth_map"] = self._load_map(dataset_dict["depth_image_path"], 'depth', image_shape) \
                if 'depth_image_path' in dataset_dict else None
            dataset_dict["ground_map"] = self._load_map(dataset_dict["ground_image_path"], 'mask', image_shape) \
                if 'ground_image_path' in dataset_dict else None

        dataset_dict["image"] = torch.as_tensor(np.ascontiguousarray(image.transpose(2, 0, 1)))
        if not self.is_train:
            return dataset_dict

        if "annotations" in dataset_dict:
            K = np.array(dataset_dict['K'])
            if self.dataset_id_to_unknown_cats is None:
                raise RuntimeError("DatasetMapper3D.dataset_id_to_unknown_cats is not set (tools/train_net.py:147 assigns it "
                                   "after building the loader; see cubercnn.data.build.dataset_id_maps)")
            unknown = self.dataset_id_to_unknown_cats[dataset_dict['dataset_id']]
            annos = [transform_instance_annotations(obj, transforms, K=K)
                     for obj in dataset_dict.pop("annotations") if obj.get("iscrowd", 0) == 0]
            instances = annotations_to_instances(annos, image_shape, unknown)
            dataset_dict["instances"] = D.filter_empty_instances(instances)
        return dataset_dict


def transform_instance_annotations(annotation, transforms, *, K):
    """dataset_mapper.py:192-247: moves the 2D box, the projected 3D centre and the 8 projected corners through the
    image transforms; a horizontal flip also mirrors the pose."""
    if isinstance(transforms, (tuple, list)):
        transforms = D.TransformList(transforms)
    box = BoxMode.convert(annotation["bbox"], annotation["bbox_mode"], BoxMode.XYXY_ABS)
    annotation["bbox"] = transforms.apply_box(np.array([box]))[0]
    annotation["bbox_mode"] = BoxMode.XYXY_ABS

    if annotation['center_cam'][2] != 0:
        proj = K @ np.array(annotation['center_cam'])
        proj[:2] = proj[:2] / proj[-1]
        annotation["center_cam_proj"] = proj.tolist()
        annotation["center_cam_proj"][0:2] = transforms.apply_coords(proj[np.newaxis][:, :2])[0].tolist()

        kps = (K @ np.array(annotation["bbox3D_cam"]).T).T
        kps[:, 0] /= kps[:, -1]
        kps[:, 1] /= kps[:, -1]
        # third column becomes the visibility flag: 1 = not visible (ignored object), 2 = visible
        kps[:, 2] = 1 if annotation['ignore'] else 2
        transforms.apply_coords(kps[:, :2])              # in place on the view
        annotation["keypoints"] = kps.tolist()

        for t in transforms:
            if isinstance(t, D.HFlipTransform):
                pose = _M1 @ np.array(annotation["pose"]) @ _M2
                annotation["pose"] = pose.tolist()
                annotation["R_cam"] = pose.tolist()
    return annotation


def annotations_to_instances(annos, image_size, unknown_categories):
    """dataset_mapper.py:250-272"""
    target = Instances(image_size)
    target.gt_classes = torch.tensor([int(o["category_id"]) for o in annos], dtype=torch.int64)
    target.gt_boxes = Boxes(np.array([BoxMode.convert(o["bbox"], o["bbox_mode"], BoxMode.XYXY_ABS) for o in annos],
                                     dtype=np.float32).reshape(-1, 4))
    target.gt_boxes3D = torch.tensor([a['center_cam_proj'] + a['dimensions'] + a['center_cam'] for a in annos],
                                     dtype=torch.float32).reshape(-1, 9)
    target.gt_poses = torch.tensor([a['pose'] for a in annos], dtype=torch.float32).reshape(-1, 3, 3)
    n = len(target.gt_classes)
    target.gt_keypoints = Keypoints(torch.tensor([a['keypoints'] for a in annos], dtype=torch.float32).reshape(-1, 8, 3))
    mask = torch.zeros(max(unknown_categories) + 1, dtype=torch.bool)
    mask[torch.tensor(list(unknown_categories))] = True
    target.gt_unknown_category_mask = mask.unsqueeze(0).repeat([n, 1])
    return target
