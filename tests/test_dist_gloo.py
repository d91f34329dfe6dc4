"""CPU, world_size 2, gloo: the data-parallel protocol of TrainStep (fused loss all-reduce, bucketed gradient
all-reduce of the flat buffer, device-side skip flag) keeps the replicas bit-identical, and a non-finite gradient
on ONE rank skips the update on BOTH.  The model runs on the float32 torch backend of oracle/cpu_backend.py
(test infrastructure); the collective logic under test is the product's."""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import cpu_backend
    cpu_backend.install()
    syn = importlib.import_module("3dod_amd.synthetic")
    modeling = importlib.import_module("3dod_amd.cubercnn.modeling")
    solver = importlib.import_module("3dod_amd.cubercnn.solver")
    d2 = importlib.import_module("3dod_amd.d2lite")
    cfg = syn.make_cfg(overrides=["MODEL.DEVICE", "cpu", "VIS_PERIOD", 0, "log", False, "SOLVER.BASE_LR", 0.01,
                                  "MODEL.RPN.PRE_NMS_TOPK_TRAIN", 200, "MODEL.RPN.POST_NMS_TOPK_TRAIN", 100,
                                  "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 64])
    torch.manual_seed(0)                       # same init on both ranks (DDP would broadcast)
    model = modeling.build_model(cfg)
    model.train()
    opt = solver.build_optimizer(cfg, model)
    dist.broadcast(opt.flat_p, 0)
    step = solver.TrainStep(cfg, model, opt, world_size=world, bucket_mb=16)
    assert len(step.buckets) > 1
    torch.manual_seed(100 + rank)              # different sampling streams / data per rank
    p0 = opt.flat_p.clone()
    with d2.EventStorage(0):
        step(syn.make_batch(1, 10 + rank, size=128, min_obj=2, max_obj=4))
        rep1 = step.report()
        p1 = opt.flat_p.clone()
        # poison one gradient on rank 1 only: both ranks must skip
        real_collect = opt.collect_grads

        def poisoned():
            real_collect()
            if rank == 1:
                opt.flat_g[123] = float("nan")
        opt.collect_grads = poisoned
        step(syn.make_batch(1, 20 + rank, size=128, min_obj=2, max_obj=4))
        rep2 = step.report()
    gathered = [torch.zeros_like(p1) for _ in range(world)]
    dist.all_gather(gathered, opt.flat_p)
    q.put((rank, bool(torch.equal(gathered[0], gathered[1])), bool(not torch.equal(p0, p1)),
           bool(torch.equal(p1, opt.flat_p)), rep1["total_loss"], rep2["iterations_explode"]))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_train_step_protocol():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res.sort()
    for rank, same, moved, skipped_kept, total, explode in res:
        assert same, "replicas diverged"
        assert moved, "first step did not update the parameters"
        assert skipped_kept, "the poisoned step was not skipped on every rank"
        assert explode == 1.0
    assert res[0][4] == res[1][4], "the fused loss all-reduce must give every rank the same total"


def test_allreduce_range_bookkeeping():
    """early (RoI-head FC weights) + late ranges of the flat gradient tile [0, n) exactly once."""
    import importlib
    b = importlib.import_module("3dod_amd.cubercnn.solver.build")
    n, bucket = 1000, 64
    for early in ([], [(0, 10)], [(990, 1000)], [(100, 300), (300, 420), (700, 701)], [(0, 1000)]):
        late = b.complement_ranges(early, n, bucket)
        cover = sorted(list(early) + late)
        assert all(e - a <= bucket for a, e in late)
        pos = 0
        for a, e in cover:
            assert a == pos and e > a, (early, cover)
            pos = e
        assert pos == n


def _data_worker(rank, world, port, root, q):
    """data path across two real ranks: samplers take rank / world size from the process group, the train stream is the
    interleaved split of one seeded permutation stream, the test shards cover the dataset once, and
    inference_on_dataset gathers every rank's records on rank 0."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    os.chdir(root)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import copy
        import itertools
        syn = importlib.import_module("3dod_amd.synthetic")
        data = importlib.import_module("3dod_amd.cubercnn.data")
        D = importlib.import_module("3dod_amd.d2lite.data")
        d2 = importlib.import_module("3dod_amd.d2lite")
        ev = importlib.import_module("3dod_amd.cubercnn.evaluation")
        cats = ["bed", "car", "chair", "sofa", "table", "truck"]
        cfg = syn.make_cfg(overrides=["MODEL.DEVICE", "cpu", "DATASETS.TRAIN", ("Synth_train",), "DATASETS.CATEGORY_NAMES", cats,
                                      "MODEL.ROI_HEADS.NUM_CLASSES", len(cats), "SOLVER.IMS_PER_BATCH", 4,
                                      "DATALOADER.NUM_WORKERS", 0, "DATALOADER.ASPECT_RATIO_GROUPING", False,
                                      "INPUT.MIN_SIZE_TRAIN", (128,), "INPUT.MAX_SIZE_TRAIN", 256, "INPUT.MIN_SIZE_TEST", 128,
                                      "INPUT.MAX_SIZE_TEST", 256, "SEED", 3])
        fs = data.get_filter_settings_from_cfg(cfg)
        omni = data.Omni3D([os.path.join("datasets", "Omni3D", "Synth_train.json")], filter_settings=copy.deepcopy(fs))
        data.register_and_store_model_metadata(omni, root, fs)
        data.simple_register("Synth_train", fs, filter_empty=True)
        meta = D.MetadataCatalog.get("omni3d_model")
        unknown, id_to_src = data.build.dataset_id_maps(omni, len(cats), meta.thing_dataset_id_to_contiguous_id)
        mapper = data.DatasetMapper3D(cfg, is_train=True)
        mapper.dataset_id_to_unknown_cats = unknown
        loader = data.build_detection_train_loader(cfg, mapper=mapper, dataset_id_to_src=id_to_src)     # rank from dist
        train_ids = [[d["image_id"] for d in b] for b in itertools.islice(iter(loader), 3)]
        assert all(len(b) == 2 for b in train_ids)                 # global 4 / 2 ranks

        class Echo(torch.nn.Module):           # a "model" that returns one fixed detection per image
            def forward(self, inputs):
                out = []
                for d in inputs:
                    inst = d2.Instances((d["height"], d["width"]))
                    inst.pred_boxes = d2.Boxes(torch.tensor([[1., 2., 30., 40.]]))
                    inst.scores = torch.tensor([0.5 + 0.001 * rank])
                    inst.pred_classes = torch.tensor([1])
                    out.append({"instances": inst})
                return out

        tl = data.build_detection_test_loader(cfg, "Synth_train", batch_size=2, num_workers=0)
        local_ids = [d["image_id"] for b in tl for d in b]
        recs = ev.inference_on_dataset(Echo(), data.build_detection_test_loader(cfg, "Synth_train", batch_size=2, num_workers=0))
        q.put((rank, train_ids, local_ids, [r["image_id"] for r in recs], [r["instances"][0]["score"] for r in recs],
               len(D.DatasetCatalog.get("Synth_train"))))
    finally:
        dist.destroy_process_group()


def test_data_path_two_ranks(tmp_path):
    syn = importlib.import_module("3dod_amd.synthetic")
    root = tmp_path / "datasets"
    root.mkdir()
    syn.make_omni3d_dataset(str(root), name="Synth_train", n_images=9, seed=4)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_data_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, tr0, loc0, got0, sc0, n_train), (_, tr1, loc1, got1, sc1, _) = res
    # one seeded permutation stream, rank r keeps elements r, r+2, ...: the first n_train draws are all different
    flat0, flat1 = [i for b in tr0 for i in b], [i for b in tr1 for i in b]
    merged = [x for pair in zip(flat0, flat1) for x in pair]
    assert len(merged) > n_train and len(set(merged[:n_train])) == n_train
    # contiguous test shards, complete, in order; rank 0 receives everything
    assert loc0 + loc1 == sorted(loc0 + loc1) and len(set(loc0) & set(loc1)) == 0
    assert got0 == loc0 + loc1 and got1 == []
    assert sc0[:len(loc0)] == [0.5] * len(loc0) and all(abs(s - 0.501) < 1e-6 for s in sc0[len(loc0):])
