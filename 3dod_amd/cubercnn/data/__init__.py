"""Omni3D data path: annotation json -> dataset dicts -> mapped per-image dicts with `Instances`
-> per-rank batches staged onto the GPU.  Mirrors the reference's `cubercnn.data` names."""
from .datasets import *  # noqa: F401,F403
from .datasets import get_version, get_filter_settings_from_cfg, is_ignore, Omni3D, COCO, load_omni3d_json, \
    simple_register, register_and_store_model_metadata, get_global_dataset_stats, save_global_dataset_stats  # noqa: F401
from .dataset_mapper import DatasetMapper3D, transform_instance_annotations, annotations_to_instances  # noqa: F401
from .build import get_detection_dataset_dicts, filter_images_with_only_crowd_annotations, \
    repeat_factors_from_category_frequency, build_detection_train_loader, build_detection_test_loader, \
    DevicePrefetcher  # noqa: F401
from .builtin import get_omni3d_categories  # noqa: F401
