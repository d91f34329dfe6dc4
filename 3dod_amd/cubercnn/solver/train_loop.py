"""The training driver around `TrainStep`: learning-rate schedule, checkpoint / resume, the retry rule and periodic
evaluation (reference: tools/train_net.py:127-333 `do_train`, cubercnn/solver/checkpoint.py, and detectron2's
build_lr_scheduler / DetectionCheckpointer [third-party, restated]).

The reference asks the host for the loss and a gradient scan EVERY iteration (`.item()`, torch.isnan(...).any()).  Here
the step keeps its divergence bookkeeping on the device; the host looks at it every `check_period` iterations (one sync),
which is where the retry rule of train_net.py:268-302 is evaluated -- collectively, so all ranks return together.
"""
import bisect
import logging
import math
import os

import torch
import torch.distributed as dist

from ...d2lite import EventStorage, JSONWriter, CommonMetricPrinter
from .build import TrainStep, build_optimizer, freeze_bn

logger = logging.getLogger(__name__)


class WarmupMultiStepLR:
    """detectron2 build_lr_scheduler(name="WarmupMultiStepLR") = LRMultiplier(WarmupParamScheduler(MultiStepParamScheduler)):
    factor(t) = gamma^(#milestones <= t), and during the first warmup_iters iterations an interpolation from
    warmup_factor * factor(0) to factor(warmup_iters).  Drives `optimizer.lr_scale` (FlatSGD multiplies every group's
    learning rate by it, like LambdaLR)."""

    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=0.001, warmup_iters=1000, warmup_method="linear",
                 max_iter=None, last_iter=-1):
        if list(milestones) != sorted(milestones):
            raise ValueError("Milestones should be a list of increasing integers. Got {}".format(milestones))
        if warmup_method not in ("constant", "linear"):
            raise ValueError("Unknown warmup method: {}".format(warmup_method))
        if max_iter is not None:                    # detectron2 drops (with a warning) steps beyond MAX_ITER
            milestones = [s for s in milestones if s <= max_iter]
        self.optimizer, self.milestones, self.gamma = optimizer, list(milestones), gamma
        self.warmup_factor, self.warmup_iters, self.warmup_method = warmup_factor, warmup_iters, warmup_method
        self.last_iter = last_iter
        self.step()

    def _plateau(self, t):
        return self.gamma ** bisect.bisect_right(self.milestones, t)

    def factor(self, t):
        if t >= self.warmup_iters or self.warmup_iters <= 0:
            return self._plateau(t)
        start, end = self.warmup_factor * self._plateau(0), self._plateau(self.warmup_iters)
        if self.warmup_method == "constant":
            return start
        alpha = t / self.warmup_iters
        return start * (1 - alpha) + end * alpha

    def step(self):
        self.last_iter += 1
        self.optimizer.lr_scale = self.factor(self.last_iter)

    def get_last_lr(self):
        return [g["lr"] * self.optimizer.lr_scale for g in self.optimizer.param_groups]

    def state_dict(self):
        return {"last_iter": self.last_iter}

    def load_state_dict(self, sd):
        self.last_iter = sd["last_iter"]
        self.optimizer.lr_scale = self.factor(self.last_iter)


class WarmupCosineLR(WarmupMultiStepLR):
    def __init__(self, optimizer, max_iter, warmup_factor=0.001, warmup_iters=1000, warmup_method="linear", end_value=0.0,
                 last_iter=-1):
        self.max_iter_, self.end_value = max_iter, end_value
        super().__init__(optimizer, [], 1.0, warmup_factor, warmup_iters, warmup_method, None, last_iter)

    def _plateau(self, t):
        w = min(max(t / self.max_iter_, 0.0), 1.0)
        return self.end_value + 0.5 * (1.0 - self.end_value) * (1 + math.cos(math.pi * w))


def build_lr_scheduler(cfg, optimizer):
    S = cfg.SOLVER
    name = S.LR_SCHEDULER_NAME
    if name == "WarmupMultiStepLR":
        return WarmupMultiStepLR(optimizer, list(S.STEPS), S.GAMMA, S.WARMUP_FACTOR, S.WARMUP_ITERS, S.WARMUP_METHOD, S.MAX_ITER)
    if name == "WarmupCosineLR":
        return WarmupCosineLR(optimizer, S.MAX_ITER, S.WARMUP_FACTOR, S.WARMUP_ITERS, S.WARMUP_METHOD)
    raise ValueError("Unknown LR scheduler: {}".format(name))


class Checkpointer:
    """model + optimizer (momentum buffer) + scheduler + the step's divergence bookkeeping + RNG state in one `.pth`;
    `last_checkpoint` names the file to resume from (fvcore Checkpointer layout).  Rank 0 writes."""

    def __init__(self, model, save_dir, optimizer=None, scheduler=None, step=None, save_to_disk=True):
        self.model, self.save_dir, self.optimizer, self.scheduler, self.step_obj = model, save_dir, optimizer, scheduler, step
        self.save_to_disk = save_to_disk

    @staticmethod
    def _dist():
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    def save(self, name, **extra):
        """called by EVERY rank (rank 0 writes): the random-generator states of all ranks are gathered into the file so
        that a resumed run continues each rank's own sampling stream instead of cloning rank 0's"""
        rank, world = self._dist()
        rng = {"cpu": torch.get_rng_state(), "cuda": torch.cuda.get_rng_state_all() if torch.cuda.is_available() else None}
        rng_by_rank = None
        if world > 1:
            rng_by_rank = [None] * world if rank == 0 else None
            dist.gather_object(rng, rng_by_rank, dst=0)
        if not self.save_dir or not self.save_to_disk:
            return None
        os.makedirs(self.save_dir, exist_ok=True)
        data = {"model": {k: v.detach().cpu() for k, v in self.model.state_dict().items()}}
        if self.optimizer is not None:
            data["optimizer"] = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v)
                                 for k, v in self.optimizer.state_dict().items()}
        if self.scheduler is not None:
            data["scheduler"] = self.scheduler.state_dict()
        if self.step_obj is not None:
            s = self.step_obj
            data["train_step"] = {"recent_loss": s.recent_loss.cpu(), "iterations_success": s.iterations_success.cpu(),
                                  "iterations_explode": s.iterations_explode.cpu()}
        data["rng"] = rng
        if rng_by_rank is not None:
            data["rng_by_rank"] = rng_by_rank
        data.update(extra)
        path = os.path.join(self.save_dir, name + ".pth")
        tmp = path + ".tmp"
        torch.save(data, tmp)
        os.replace(tmp, path)                        # a crash mid-write never leaves a truncated checkpoint behind
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
            f.write(os.path.basename(path))
        return path

    def has_checkpoint(self):
        return bool(self.save_dir) and os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

    def get_checkpoint_file(self):
        with open(os.path.join(self.save_dir, "last_checkpoint")) as f:
            return os.path.join(self.save_dir, f.read().strip())

    def load(self, path, checkpointables=None):
        """checkpointables=[]: the model only (MODEL.WEIGHTS_PRETRAIN, train_net.py:151-154).  Keys of the checkpoint that
        the model does not have and parameters the checkpoint does not cover are LOGGED (as detectron2's checkpointer does,
        [third-party]); a checkpoint that matches no parameter at all is an error, not a silent random initialisation."""
        if not path:
            return {}
        data = torch.load(path, map_location="cpu", weights_only=True)
        sd = data.pop("model") if "model" in data else data
        own = self.model.state_dict()
        shape_bad = [k for k, v in sd.items() if k in own and hasattr(v, "shape") and tuple(v.shape) != tuple(own[k].shape)]
        if shape_bad:
            for k in shape_bad:
                logger.warning("checkpoint %s: shape of '%s' is %s, the model has %s -- skipped", path, k,
                               tuple(sd[k].shape), tuple(own[k].shape))
            sd = {k: v for k, v in sd.items() if k not in shape_bad}
        res = self.model.load_state_dict(sd, strict=False)
        self.incompatible = {"missing": list(res.missing_keys), "unexpected": list(res.unexpected_keys)}     # (mis-shaped keys count as missing)
        if res.missing_keys:
            logger.warning("checkpoint %s: %d model keys not in the checkpoint (left at their initial values): %s", path,
                           len(res.missing_keys), ", ".join(res.missing_keys[:20]) + (" ..." if len(res.missing_keys) > 20 else ""))
        if res.unexpected_keys:
            logger.warning("checkpoint %s: %d checkpoint keys the model does not have (ignored): %s", path,
                           len(res.unexpected_keys), ", ".join(res.unexpected_keys[:20]) + (" ..." if len(res.unexpected_keys) > 20 else ""))
        if len(own) and len(res.missing_keys) >= len(own):
            raise RuntimeError(f"checkpoint {path} matches none of the model's {len(own)} state-dict keys "
                               f"(first checkpoint keys: {list(sd)[:5]})")
        want = lambda k: checkpointables is None or k in checkpointables
        dev = next(self.model.parameters()).device
        if want("optimizer") and self.optimizer is not None and "optimizer" in data:
            self.optimizer.load_state_dict({k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in data["optimizer"].items()})
        if want("scheduler") and self.scheduler is not None and "scheduler" in data:
            self.scheduler.load_state_dict(data["scheduler"])
        if want("train_step") and self.step_obj is not None and "train_step" in data:
            s, t = self.step_obj, data["train_step"]
            s.recent_loss.copy_(t["recent_loss"])
            s.iterations_success.copy_(t["iterations_success"])
            s.iterations_explode.copy_(t["iterations_explode"])
        if want("rng") and "rng" in data:
            rank, world = self._dist()
            by_rank = data.get("rng_by_rank")
            if by_rank is not None and len(by_rank) == world:
                mine = by_rank[rank]                   # every rank continues ITS OWN stream
            else:
                mine = data["rng"]
            torch.set_rng_state(mine["cpu"])
            if mine["cuda"] is not None and torch.cuda.is_available():
                torch.cuda.set_rng_state_all(mine["cuda"])
            if (by_rank is None or len(by_rank) != world) and rank > 0:
                # only rank 0's state is on file (single-rank checkpoint, or another world size): ranks must not share a
                # stream -- re-seed the others from it, deterministically per rank
                base = int(torch.initial_seed())
                torch.manual_seed(base + 7919 * rank)
        return data

    def resume_or_load(self, path, resume=True):
        if resume and self.has_checkpoint():
            return self.load(self.get_checkpoint_file())
        return self.load(path, checkpointables=[])


class PeriodicCheckpointerOnlyOne:
    """cubercnn/solver/checkpoint.py:5-28: one rolling `model_recent` every `period` iterations and `model_final`"""

    def __init__(self, checkpointer, period, max_iter=None, file_prefix="model"):
        self.checkpointer, self.period, self.max_iter, self.file_prefix = checkpointer, int(period), max_iter, file_prefix

    def step(self, iteration, **kwargs):
        iteration = int(iteration)
        if (iteration + 1) % self.period == 0:
            self.checkpointer.save("{}_recent".format(self.file_prefix), iteration=iteration, **kwargs)
        if self.max_iter is not None and iteration >= self.max_iter - 1:
            self.checkpointer.save(f"{self.file_prefix}_final", iteration=iteration, **kwargs)


def make_train_step(cfg, model, optimizer, world_size=1):
    """The step object `do_train` runs (and `bench.py` times): `TrainStep` with the dense region (preprocess, trunk, FPN,
    RPN head: forward and backward) replayed from HIP graphs, one captured region per image-batch shape met, captured on
    first sight and kept for the CR_GRAPH_SHAPES (8) most recently used shapes -- the eager dense region is ~500 launches
    and bound by the host's launch rate.  Data-parallel runs capture the backward as two graphs so that the first
    segment's gradient all-reduce travels under the second (graphed.GraphedDense).  CR_GRAPHS=none: eager launches.
    Batches whose images differ in size run eagerly (the padded border is masked after normalisation)."""
    step = TrainStep(cfg, model, optimizer, world_size=world_size)
    mode = os.environ.get("CR_GRAPHS", "dense")
    dev = optimizer.flat_p.device
    if mode != "none" and dev.type == "cuda" and getattr(model, "dense_train", False) and hasattr(model, "enable_graphs"):
        model.enable_graphs(None, split_backward=world_size > 1 and os.environ.get("CR_BWD_SPLIT", "1") != "0",
                            max_shapes=int(os.environ.get("CR_GRAPH_SHAPES", "8")))
    return step


def do_train(cfg, model, data_loader, resume=False, world_size=None, rank=None, do_test=None, check_period=20):
    """tools/train_net.py:127-333.  `data_loader`: iterable of per-rank batches (build_detection_train_loader, ideally
    wrapped in DevicePrefetcher).  Returns True on success, False when the run should be restarted from the last
    checkpoint because too many iterations were skipped (the caller loops over attempts like train_net.py:452-493)."""
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    max_iter = cfg.SOLVER.MAX_ITER
    do_eval = cfg.TEST.EVAL_PERIOD > 0 and do_test is not None
    model.train()
    optimizer = build_optimizer(cfg, model)
    scheduler = build_lr_scheduler(cfg, optimizer)
    step = make_train_step(cfg, model, optimizer, world_size=world_size)
    checkpointer = Checkpointer(model, cfg.OUTPUT_DIR, optimizer=optimizer, scheduler=scheduler, step=step,
                                save_to_disk=rank == 0)
    periodic = PeriodicCheckpointerOnlyOne(checkpointer, cfg.SOLVER.CHECKPOINT_PERIOD, max_iter=max_iter)
    if cfg.MODEL.WEIGHTS_PRETRAIN != '':
        checkpointer.load(cfg.MODEL.WEIGHTS_PRETRAIN, checkpointables=[])
    start_iter = checkpointer.resume_or_load(cfg.MODEL.get("WEIGHTS", ""), resume=resume).get("iteration", -1) + 1
    if world_size > 1:
        dist.broadcast(optimizer.flat_p, 0)           # DDP's wrap-time parameter broadcast
    logger.info("Starting training from iteration {}".format(start_iter))
    if not cfg.MODEL.USE_BN:
        freeze_bn(model)
    dev = optimizer.flat_p.device
    iteration = start_iter
    data_iter = iter(data_loader)
    # default_writers of the reference (train_net.py:140): stdout + OUTPUT_DIR/metrics.json, on rank 0
    writers = [CommonMetricPrinter(max_iter), JSONWriter(os.path.join(cfg.OUTPUT_DIR, "metrics.json"))] \
        if rank == 0 and cfg.OUTPUT_DIR else []
    with EventStorage(start_iter) as storage:
        while iteration < max_iter:
            storage.iter = iteration
            step(next(data_iter))
            scheduler.step()
            last = iteration == max_iter - 1
            skipped_now = False
            if (iteration + 1) % check_period == 0 or last or (do_eval and (iteration + 1) % cfg.TEST.EVAL_PERIOD == 0) \
                    or (iteration + 1) % cfg.SOLVER.CHECKPOINT_PERIOD == 0:
                rep = step.report()                  # the one host sync of these iterations
                ok, bad = rep["iterations_success"], rep["iterations_explode"]
                total = max(ok + bad, 1.0)
                skipped_now = bool(step.last["skipped"].item())
                storage.put_scalars(lr=scheduler.get_last_lr()[0], **{k: v for k, v in rep.items() if not k.startswith("iterations_")})
                for w in writers:
                    w.write(storage)
                retry = torch.tensor(float(bad / total >= cfg.MODEL.STABILIZE > 0 and total > cfg.SOLVER.CHECKPOINT_PERIOD / 2),
                                     device=dev)
                if world_size > 1:
                    dist.all_reduce(retry)
                if float(retry) > 0:
                    logger.warning('!! Restarting training at {} iters. Exploding loss {:d}% of iters !!'.format(
                        iteration, int(100 * bad / total)))
                    return False
                if do_eval and not skipped_now and (iteration + 1) % cfg.TEST.EVAL_PERIOD == 0 and not last:
                    do_test(cfg, model, iteration=iteration + 1, storage=storage)
                    model.train()
                    if not cfg.MODEL.USE_BN:
                        freeze_bn(model)
                # no checkpoint while the model may be diverging (train_net.py:323-326)
                if not skipped_now and bad / total < 0.5 * cfg.MODEL.STABILIZE or cfg.MODEL.STABILIZE <= 0:
                    periodic.step(iteration)
            iteration += 1
    for w in writers:
        if hasattr(w, "close"):
            w.close()
    return True
