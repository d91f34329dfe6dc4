#!/usr/bin/env python
"""Training / evaluation driver on the MI355X path -- the counterpart of the reference's tools/train_net.py
(main :356-495, do_train :127-333, do_test :65-125) built on this repo's packages only:

    python tools/train_net.py --config-file configs/Base_Omni3D.yaml OUTPUT_DIR output/run1 [KEY VALUE ...]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_net.py --config-file ... (one rank per GPU)
    python tools/train_net.py --config-file ... --eval-only MODEL.WEIGHTS output/run1/model_final.pth

Datasets are Omni3D json files under datasets/Omni3D/<name>.json (3dod_amd.synthetic.make_omni3d_dataset writes a small
one in that format).  Up to MAX_TRAINING_ATTEMPTS restarts from the last checkpoint when too many steps were skipped.
"""
import argparse
import importlib
import logging
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MAX_TRAINING_ATTEMPTS = 10
logger = logging.getLogger("cubercnn")


def _mods():
    m = lambda n: importlib.import_module("3dod_amd." + n)
    return m("synthetic"), m("cubercnn.data"), m("cubercnn.evaluation"), m("cubercnn.util"), m("cubercnn.modeling"), \
        m("cubercnn.solver"), m("d2lite.data")


def do_test(cfg, model, iteration='final', storage=None):
    """train_net.py:65-125: every TEST dataset through the model on this rank's shard, records gathered on rank 0, AP2D /
    AP3D per dataset and pooled."""
    syn, data, ev, util, modeling, solver, D = _mods()
    fs = data.get_filter_settings_from_cfg(cfg)
    fs['category_names'] = D.MetadataCatalog.get('omni3d_model').thing_classes
    names = list(cfg.DATASETS.TEST)
    rank0 = not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
    helper = ev.Omni3DEvaluationHelper(names, fs, os.path.join(cfg.OUTPUT_DIR, "inference", f"iter_{iteration}"),
                                       iter_label=iteration, only_2d=cfg.MODEL.ROI_CUBE_HEAD.LOSS_W_3D == 0.0) if rank0 else None
    for name in names:
        if name not in D.DatasetCatalog:
            data.simple_register(name, fs, filter_empty=False)
        preds = ev.inference_on_dataset(model, data.build_detection_test_loader(cfg, name, num_workers=0))
        if rank0:
            helper.add_predictions(name, preds)
            helper.evaluate(name)
    if not rank0:
        return {}
    analysis, omni = helper.summarize_all()
    for k, v in analysis.items():
        logger.info("%s  AP2D %.2f  AP3D %.2f", k, v["AP2D"], v["AP3D"])
    return analysis


def setup(args):
    syn = _mods()[0]
    cfg = syn.make_cfg(args.config_file, overrides=args.opts)
    if torch.cuda.is_available() and "MODEL.DEVICE" not in args.opts:
        cfg.MODEL.DEVICE = "cuda:%d" % int(os.environ.get("LOCAL_RANK", 0))
    torch.manual_seed(int(cfg.SEED) if int(cfg.get("SEED", -1)) >= 0 else 0)
    return cfg


def main(args):
    syn, data, ev, util, modeling, solver, D = _mods()
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world > 1 and not dist.is_initialized():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        dist.init_process_group("nccl")                        # RCCL on ROCm
    cfg = setup(args)
    os.makedirs(cfg.OUTPUT_DIR, exist_ok=True)
    fs = data.get_filter_settings_from_cfg(cfg)
    for name in cfg.DATASETS.TRAIN:
        if name not in D.DatasetCatalog:
            data.simple_register(name, fs, filter_empty=True)
    root = os.path.join('datasets', 'Omni3D')
    omni = data.Omni3D([os.path.join(root, n + '.json') for n in cfg.DATASETS.TRAIN], filter_settings=fs)
    data.register_and_store_model_metadata(omni, cfg.OUTPUT_DIR, fs)
    meta = D.MetadataCatalog.get('omni3d_model')
    cfg.MODEL.ROI_HEADS.NUM_CLASSES = len(meta.thing_classes)
    unknown, id_to_src = data.build.dataset_id_maps(omni, cfg.MODEL.ROI_HEADS.NUM_CLASSES, meta.thing_dataset_id_to_contiguous_id)
    priors = util.compute_priors(cfg, omni)

    attempts = MAX_TRAINING_ATTEMPTS
    while attempts > 0:
        model = modeling.build_model(cfg, priors=priors)
        if args.eval_only:
            solver.Checkpointer(model, cfg.OUTPUT_DIR).resume_or_load(cfg.MODEL.get("WEIGHTS", ""), resume=args.resume)
            return do_test(cfg, model.eval())
        mapper = data.DatasetMapper3D(cfg, is_train=True)
        mapper.dataset_id_to_unknown_cats = unknown
        loader = data.build_detection_train_loader(cfg, mapper=mapper, dataset_id_to_src=id_to_src)
        feed = data.DevicePrefetcher(loader, cfg.MODEL.DEVICE)
        # a restart resumes from the checkpoint written before the divergence (train_net.py:483-490)
        if solver.do_train(cfg, model, feed, resume=args.resume or attempts < MAX_TRAINING_ATTEMPTS, do_test=do_test):
            return do_test(cfg, model.eval())
        attempts -= 1
        del model
    raise RuntimeError('Training failed')


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-file", default=os.path.join(ROOT, "configs", "Base_Omni3D.yaml"))
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--eval-only", action="store_true")
    ap.add_argument("opts", nargs=argparse.REMAINDER, default=[])
    a = ap.parse_args()

    def _val(v):
        import ast
        try:
            return ast.literal_eval(v)
        except Exception:
            return v
    a.opts = [(_val(v) if i % 2 else v) for i, v in enumerate(a.opts)]
    logging.basicConfig(level=logging.INFO)
    print(main(a))
