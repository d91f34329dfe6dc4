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
