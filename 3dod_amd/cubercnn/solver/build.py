"""Optimizer + training step of the reference (cubercnn/solver/build.py:6-76, tools/train_net.py:184-304) for
one process per GPU:

  * all parameters / gradients / momentum live in three flat float32 buffers (parameters are views), ordered
    so that the weight-decayed group is one contiguous range and the no-decay group (BatchNorm affine with
    WEIGHT_DECAY_NORM, priors) another -> the SGD-momentum update is two fused kernel launches;
  * the gradient all-reduce (DistributedDataParallel in train_net.py:477-480) is a bucketed RCCL all-reduce
    of the flat gradient, issued on a side stream so buckets overlap; the loss dict, the divergence flag and
    the non-finite flag (three collectives + three barriers in the reference) ride in ONE small all-reduce;
  * the loss-divergence guard (rolling mean x4, train_net.py:202-220) and the non-finite gradient scan
    (:233-244) are evaluated on the device; the update is skipped on the device (no host sync in the step).
"""
from typing import Any, Dict, List

import torch
import torch.distributed as dist

from ... import hipops as ops
from ..modeling.graphed import GraphOwner as _GraphOwner

NORM_TYPES = (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d, torch.nn.SyncBatchNorm,
              torch.nn.GroupNorm, torch.nn.InstanceNorm1d, torch.nn.InstanceNorm2d, torch.nn.InstanceNorm3d,
              torch.nn.LayerNorm, torch.nn.LocalResponseNorm)
NO_DECAY_KEYS = ('priors_dims_per_cat', 'priors_z_scales', 'priors_z_stats')


def freeze_bn(network):
    """solver/build.py:71-76."""
    for _, module in network.named_modules():
        if isinstance(module, torch.nn.BatchNorm2d):
            module.eval()
            module.track_running_stats = False


def _param_groups(cfg, model):
    """same per-parameter (lr, weight_decay) rule as solver/build.py:21-47."""
    memo, groups = set(), []
    for module in model.modules():
        for key, value in module.named_parameters(recurse=False):
            if not value.requires_grad or value in memo:
                continue
            memo.add(value)
            lr, wd = cfg.SOLVER.BASE_LR, cfg.SOLVER.WEIGHT_DECAY
            if isinstance(module, NORM_TYPES) and cfg.SOLVER.WEIGHT_DECAY_NORM is not None:
                wd = cfg.SOLVER.WEIGHT_DECAY_NORM
            elif key == "bias":
                if cfg.SOLVER.BIAS_LR_FACTOR is not None:
                    lr = cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR
                if cfg.SOLVER.WEIGHT_DECAY_BIAS is not None:
                    wd = cfg.SOLVER.WEIGHT_DECAY_BIAS
            if key in NO_DECAY_KEYS:
                wd = 0.0
            groups.append((value, float(lr), float(wd)))
    return groups


class FlatSGD:
    """torch.optim.SGD(momentum) semantics on flat buffers + fused kernels (cr_sgd_step)."""

    def __init__(self, groups, momentum, nesterov=False):
        self.momentum = float(momentum)
        self.nesterov = bool(nesterov)
        self.clip = None                                        # set_gradient_clipping
        # order by (lr, wd) so that equal-hyperparameter parameters are contiguous
        keys = sorted({(lr, wd) for _, lr, wd in groups})
        ordered = [(p, lr, wd) for k in keys for (p, lr, wd) in groups if (lr, wd) == k]
        dev = ordered[0][0].device
        pad = lambda n: (n + 3) // 4 * 4                       # 16-B aligned segments
        total = sum(pad(p.numel()) for p, _, _ in ordered)
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.segments = []                                      # (start, end, lr, wd)
        self.params = []
        off = 0
        cur = None
        for p, lr, wd in ordered:
            n = p.numel()
            cl = p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous()
            src = p.data.permute(0, 2, 3, 1).contiguous().view(-1) if cl else p.data.contiguous().view(-1)
            self.flat_p[off:off + n].copy_(src)

            def view_of(buf):
                v = buf[off:off + n]
                if cl:
                    K, C, R, S = p.shape
                    return v.view(K, R, S, C).permute(0, 3, 1, 2)       # logical KCRS over physical KRSC
                return v.view(p.shape)
            p.data = view_of(self.flat_p)
            p.grad = None
            p._cr_grad = view_of(self.flat_g)      # kernels accumulate straight into the flat gradient
            self.params.append(p)
            if cur is not None and tuple(cur[2:]) == (lr, wd):
                cur[1] = off + pad(n)
            else:
                cur = [off, off + pad(n), lr, wd]
                self.segments.append(cur)
            off += pad(n)
        self.param_groups = [{"lr": s[2], "weight_decay": s[3], "range": (s[0], s[1])} for s in self.segments]
        # the schedule's current factor lives on the device too: the update kernels read it there, so an update captured
        # in a HIP graph (GraphedTrainStep) follows warm-up / multi-step schedules driven through `lr_scale`
        self._lr_scale = 1.0
        self._lr_dev = torch.ones(1, dtype=torch.float32, device=dev)
        self._lr_dev_value = 1.0

    @property
    def lr_scale(self):
        return self._lr_scale

    @lr_scale.setter
    def lr_scale(self, v):
        self._lr_scale = float(v)

    def refresh_lr(self):
        """push the host-side schedule factor to the device scalar (a one-float fill, only when it changed; never inside
        a graph capture: the captured kernels read the scalar, the host refreshes it before each replay)"""
        if self._lr_dev_value != self._lr_scale and not (self._lr_dev.is_cuda and torch.cuda.is_current_stream_capturing()):
            self._lr_dev.fill_(self._lr_scale)
            self._lr_dev_value = self._lr_scale

    def enable_weight_bank(self):
        """one-launch-per-step recast / transpose of every conv weight (hipops.WeightBank); training-step objects call this.
        One bank PER PRECISION MODE is kept alive for the optimizer's life: captured graphs (GraphedDense, GraphedTrainStep)
        have the bank's buffer addresses baked in, so going fp32 -> bf16 -> fp32 re-attaches the same fp32 bank instead of
        freeing and re-allocating it under the graphs."""
        if not (self.flat_p.is_cuda and hasattr(ops, "WeightBank")):
            return getattr(self, "weight_bank", None)
        banks = self.__dict__.setdefault("_banks", {})
        mode = ops.precision()
        bank = banks.get(mode)
        if bank is None:
            convs = [p for p in self.params if p.dim() == 4 and p.shape[2] == p.shape[3] and
                     p.is_contiguous(memory_format=torch.channels_last)]
            if not convs:
                return None
            bank = banks[mode] = ops.WeightBank(convs, self.flat_p)     # in the precision mode of the process
            ops.bump_weight_epoch()
        elif getattr(self, "weight_bank", None) is not bank:
            bank.attach()
            ops.bump_weight_epoch()
        self.weight_bank = bank
        return bank

    def zero_grad(self):
        self.flat_g.zero_()
        for p in self.params:
            p.grad = None

    def collect_grads(self):
        """gradients that arrived through plain autograd (Linear layers, biases) are added into the flat buffer
        with one multi-tensor launch; those written by the conv/BN kernels are already there."""
        views, grads = [], []
        for p in self.params:
            if p.grad is not None:
                views.append(p._cr_grad)
                grads.append(p.grad)
        if views:
            torch._foreach_add_(views, grads)
        for p in self.params:
            p.grad = None

    def set_gradient_clipping(self, clip_type, clip_value, norm_type=2.0):
        """SOLVER.CLIP_GRADIENTS (detectron2 maybe_add_gradient_clipping, solver/build.py:68): 'value' clamps every gradient
        element, 'norm' rescales every PARAMETER's gradient on its own to at most clip_value in the norm_type norm -- applied
        to the flat gradient right before the update (after the all-reduce), one / two launches for the whole model"""
        if clip_type not in ("value", "norm"):
            raise ValueError(f"SOLVER.CLIP_GRADIENTS.CLIP_TYPE must be 'value' or 'norm', got {clip_type!r}")
        self.clip = (clip_type, float(clip_value), float(norm_type))
        if clip_type == "norm" and not hasattr(self, "_clip_starts"):
            dev = self.flat_g.device
            base = self.flat_g.data_ptr()
            st = [(p._cr_grad.data_ptr() - base) // 4 for p in self.params]
            self._clip_starts = torch.tensor(st, dtype=torch.int64, device=dev)
            self._clip_counts = torch.tensor([p.numel() for p in self.params], dtype=torch.int64, device=dev)
            self._clip_partial = torch.empty(len(st) * 16, dtype=torch.float32, device=dev)

    def clip_gradients(self, grad_scale=1.0):
        """returns the grad_scale the update has to use afterwards (1.0 when clipping folded it in)"""
        if self.clip is None:
            return grad_scale
        kind, value, norm_type = self.clip
        if kind == "value":
            ops.grad_clip_value(self.flat_g, value, grad_scale)
        else:
            ops.grad_clip_norm(self.flat_g, self._clip_starts, self._clip_counts, value, norm_type, grad_scale, self._clip_partial)
        return 1.0

    def step(self, skip_flag=None, grad_scale=1.0):
        self.refresh_lr()
        grad_scale = self.clip_gradients(grad_scale)
        for (a, b, lr, wd) in self.segments:
            ops.sgd_step(self.flat_p[a:b], self.flat_g[a:b], self.flat_m[a:b], lr, self.momentum, wd,
                         grad_scale, skip_flag, lr_scale_dev=self._lr_dev, nesterov=self.nesterov)
        ops.bump_weight_epoch()

    def state_dict(self):
        return {"momentum_buffer": self.flat_m, "lr_scale": self.lr_scale}

    def load_state_dict(self, sd):
        self.flat_m.copy_(sd["momentum_buffer"])
        self.lr_scale = sd.get("lr_scale", 1.0)
        self.refresh_lr()


class FlatAdam(FlatSGD):
    """torch.optim.Adam / AdamW (optionally amsgrad) on the flat buffers of FlatSGD: `flat_m` is exp_avg, `flat_v` exp_avg_sq,
    `flat_vmax` the amsgrad maximum; betas (0.9, 0.999) are torch's defaults, eps is the caller's (the reference passes 1e-2,
    solver/build.py:57-64).  The count of applied updates lives on the device (a skipped step does not advance it)."""

    def __init__(self, groups, eps=1e-2, betas=(0.9, 0.999), decoupled=False, amsgrad=False):
        super().__init__(groups, momentum=0.0)
        self.eps, self.betas, self.decoupled, self.amsgrad = float(eps), (float(betas[0]), float(betas[1])), bool(decoupled), bool(amsgrad)
        self.flat_v = torch.zeros_like(self.flat_m)
        self.flat_vmax = torch.zeros_like(self.flat_m) if amsgrad else None
        self.step_dev = torch.zeros(1, dtype=torch.float32, device=self.flat_p.device)

    def step(self, skip_flag=None, grad_scale=1.0):
        self.refresh_lr()
        grad_scale = self.clip_gradients(grad_scale)
        ops.adam_tick(self.step_dev, skip_flag)
        for (a, b, lr, wd) in self.segments:
            ops.adam_step(self.flat_p[a:b], self.flat_g[a:b], self.flat_m[a:b], self.flat_v[a:b],
                          None if self.flat_vmax is None else self.flat_vmax[a:b], lr, self.betas[0], self.betas[1], self.eps, wd,
                          self.step_dev, grad_scale, skip_flag, lr_scale_dev=self._lr_dev, decoupled=self.decoupled)
        ops.bump_weight_epoch()

    def state_dict(self):
        sd = {"exp_avg": self.flat_m, "exp_avg_sq": self.flat_v, "step": self.step_dev, "lr_scale": self.lr_scale}
        if self.flat_vmax is not None:
            sd["max_exp_avg_sq"] = self.flat_vmax
        return sd

    def load_state_dict(self, sd):
        self.flat_m.copy_(sd["exp_avg"])
        self.flat_v.copy_(sd["exp_avg_sq"])
        self.step_dev.copy_(sd["step"])
        if self.flat_vmax is not None:
            self.flat_vmax.copy_(sd["max_exp_avg_sq"])
        self.lr_scale = sd.get("lr_scale", 1.0)
        self.refresh_lr()


def build_optimizer(cfg, model):
    """solver/build.py:6-69: 'sgd' (the reference's default and every shipped config; SOLVER.NESTEROV honoured), 'adam',
    'adam+amsgrad', 'adamw', 'adamw+amsgrad' (eps 1e-2 as there); anything else raises like the reference.
    SOLVER.CLIP_GRADIENTS (detectron2 maybe_add_gradient_clipping [third-party], :68; off in every shipped config): 'value' and
    per-parameter 'norm' clipping on the flat gradient (cr_grad_clip_value / cr_grad_clip_norm)."""
    groups = _param_groups(cfg, model)
    t = cfg.SOLVER.TYPE
    if t == 'sgd':
        opt = FlatSGD(groups, cfg.SOLVER.MOMENTUM, cfg.SOLVER.NESTEROV)
    elif t in ('adam', 'adam+amsgrad', 'adamw', 'adamw+amsgrad'):
        opt = FlatAdam(groups, eps=1e-02, decoupled=t.startswith('adamw'), amsgrad=t.endswith('+amsgrad'))
    else:
        raise ValueError('{} is not supported as an optimizer.'.format(t))
    cg = cfg.SOLVER.get("CLIP_GRADIENTS", None)
    if cg is not None and cg.ENABLED:
        opt.set_gradient_clipping(str(cg.CLIP_TYPE).lower(), cg.CLIP_VALUE, cg.NORM_TYPE)
    return opt


def early_allreduce_ranges(model, optimizer):
    """[start, end) ranges of the flat gradient holding the FC weights of the RoI heads that carry gradient sinks."""
    rh = getattr(model, "roi_heads", None)
    if rh is None:
        return []
    base = optimizer.flat_g.data_ptr()
    out = []
    for m in list(getattr(getattr(rh, "box_head", None), "fcs", [])) + \
            [l for l in getattr(getattr(rh, "cube_head", None), "feature_generator", []) if isinstance(l, torch.nn.Linear)]:
        sink = ops.grad_sink(m.weight) if hasattr(ops, "grad_sink") else None
        if sink is not None and sink.is_contiguous():
            a = (sink.data_ptr() - base) // 4
            out.append((a, a + sink.numel()))
    out.sort()
    merged = []
    for a, b in out:
        if merged and a <= merged[-1][1]:
            merged[-1] = (merged[-1][0], max(b, merged[-1][1]))
        else:
            merged.append((a, b))
    return merged


def param_ranges(params, optimizer):
    """merged [start, end) ranges of the flat gradient covered by the sinks of `params` (16-B padded segments, FlatSGD)"""
    out = []
    for p in params:
        sink = ops.grad_sink(p) if hasattr(ops, "grad_sink") else None
        if sink is not None and sink.untyped_storage().data_ptr() == optimizer.flat_g.untyped_storage().data_ptr():
            a = sink.storage_offset()
            out.append((a, a + (p.numel() + 3) // 4 * 4))
    out.sort()
    merged = []
    for a, b in out:
        if merged and a <= merged[-1][1]:
            merged[-1] = (merged[-1][0], max(b, merged[-1][1]))
        else:
            merged.append((a, b))
    n = optimizer.flat_g.numel()
    return [(a, min(b, n)) for a, b in merged]


def complement_ranges(ranges, n, bucket):
    """[0,n) minus `ranges`, chopped into buckets of at most `bucket` elements, last parameters first."""
    out, pos = [], 0
    for a, b in list(ranges) + [(n, n)]:
        while pos < a:
            e = min(a, pos + bucket)
            out.append((pos, e))
            pos = e
        pos = max(pos, b)
    return out[::-1]


class TrainStep:
    """One iteration of do_train (tools/train_net.py:184-304) without host round trips."""
    TOLERANCE = 4.0
    GAMMA = 0.02

    def __init__(self, cfg, model, optimizer, world_size=1, bucket_mb=32, force_comm=False):
        self.model, self.opt = model, optimizer
        optimizer.enable_weight_bank()
        self.world = world_size
        self.stabilize = cfg.MODEL.STABILIZE > 0
        dev = optimizer.flat_p.device
        self.recent_loss = torch.full((), float("nan"), device=dev)
        self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self.iterations_success = torch.zeros((), device=dev)
        self.iterations_explode = torch.zeros((), device=dev)
        n = optimizer.flat_g.numel()
        be = max(1, int(bucket_mb * (1 << 20) / 4))
        self.buckets = [(i, min(i + be, n)) for i in range(0, n, be)][::-1]     # last-used params first
        self.force_comm = force_comm         # tests: run the multi-rank protocol on a 1-rank process group
        self.comm_stream = torch.cuda.Stream(device=dev) if ((world_size > 1 or force_comm) and dev.type == 'cuda') else None
        # ranges of the flat gradient that are final as soon as the RoI heads' backward is done (the big FC weights of
        # both heads: their kernels write straight into the flat gradient): all-reduced on the side stream WHILE the
        # captured trunk/FPN/RPN backward graph runs; the rest goes after backward.
        self.early_ranges = early_allreduce_ranges(model, optimizer)
        self.late_ranges = complement_ranges(self.early_ranges, n, be)
        self._early_done = False
        self.mid_ranges, self._mid_whole, self._mid_done, self._bucket = [], [], False, be
        self.last = {}

    def _mid_allreduce(self):
        """between the two captured backward segments (GraphedDense.split): the gradients of the RPN head, the FPN and the
        trunk's last level are final; their all-reduce runs on the communication stream under the second segment"""
        if self.comm_stream is None or not self.mid_ranges:
            return
        self.comm_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm_stream):
            for a, b in self.mid_ranges:
                dist.all_reduce(self.opt.flat_g[a:b])
        self._mid_done = True

    def _rest_ranges(self):
        """what the phases that ran in this step's backward have NOT all-reduced yet, in buckets (cached per combination)"""
        key = (self._early_done, self._mid_done)
        cache = self.__dict__.setdefault("_rest_cache", {})
        if key not in cache:
            done = (list(self.early_ranges) if self._early_done else []) + (list(self._mid_whole) if self._mid_done else [])
            cache[key] = complement_ranges(sorted(done), self.opt.flat_g.numel(), self._bucket) if done else self.buckets
        return cache[key]

    def _early_allreduce(self):
        if self.comm_stream is None or not self.early_ranges:
            return
        self.comm_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm_stream):
            for a, b in self.early_ranges:
                dist.all_reduce(self.opt.flat_g[a:b])
        self._early_done = True

    def __call__(self, data):
        opt, world = self.opt, self.world
        loss_dict = self.model(data)
        keys = sorted(loss_dict.keys())
        vals = torch.stack([loss_dict[k].float() for k in keys])
        losses = vals.sum()
        # ---- fused small all-reduce: loss terms (train_net.py:196 allreduce_dict).  It runs on the communication stream
        # UNDER the backward pass: backward is taken on the unclipped local loss -- the reference clips the loss to [0,1]
        # when it diverges (train_net.py:212) but then discards that step's gradients (:259-261), so the clipped backward
        # and the skipped update give the same parameters -- and the divergence decision is made after backward.
        red = vals.detach().clone()
        if world > 1:
            if self.comm_stream is not None:
                self.comm_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.comm_stream):
                    dist.all_reduce(red)
                red.record_stream(self.comm_stream)
            else:
                dist.all_reduce(red)
        opt.zero_grad()
        g = getattr(self.model, "_graphed", None)
        if g is not None and self.comm_stream is not None and g.pre_bwd is None:
            g.pre_bwd = self._early_allreduce
        if g is not None and self.comm_stream is not None and getattr(g, "bwd_graph2", None) is not None and g.mid_bwd is None:
            early = set(self.early_ranges)
            mid = [r for r in param_ranges(g.segment_params(True), opt) if r not in early]
            # bucket-sized pieces, like the other phases
            self.mid_ranges = [(x, min(x + self._bucket, b)) for a, b in mid for x in range(a, b, self._bucket)]
            self._mid_whole = mid
            self.late_after_mid = complement_ranges(sorted(self.early_ranges + mid), opt.flat_g.numel(), self._bucket)
            self.__dict__.pop("_rest_cache", None)
            g.mid_bwd = self._mid_allreduce
        self._early_done = self._mid_done = False
        losses.backward()
        opt.collect_grads()
        # ---- gradient all-reduce (DDP, train_net.py:477-480), bucketed, on a side stream; the RoI heads' part was
        # started before the trunk's backward graph (see _early_allreduce) when the dense-region graphs are active
        if self.comm_stream is not None and (world > 1 or self.force_comm):
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                for a, b in self._rest_ranges():
                    dist.all_reduce(opt.flat_g[a:b])
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        elif world > 1:                      # gloo / CPU rehearsal of the same protocol
            for a, b in self.buckets:
                dist.all_reduce(opt.flat_g[a:b])
        # ---- divergence guard (rolling mean x4, train_net.py:202-220) on the reduced loss, on the device
        losses_reduced = torch.empty((), dtype=torch.float32, device=red.device)
        ops.loss_guard(red, 1.0 / world, red, losses_reduced, self.recent_loss, self.stabilize, self.TOLERANCE, self.GAMMA,
                       self.flag)
        # ---- non-finite scan of the (averaged) gradient + skip flag, all on device
        if self.stabilize:
            ops.nonfinite_flag(opt.flat_g, self.flag)
        opt.step(skip_flag=self.flag, grad_scale=1.0 / world)
        ops.step_counters(self.flag, self.iterations_explode, self.iterations_success)
        self.last = {"keys": keys, "values": red, "total": losses_reduced, "skipped": self.flag}
        return self.last

    def report(self):
        """host-side view (one sync) for logging / the retry rule of train_net.py:268-302."""
        d = {k: float(v) for k, v in zip(self.last["keys"], self.last["values"].tolist())}
        d["total_loss"] = float(self.last["total"])
        d["iterations_explode"] = float(self.iterations_explode)
        d["iterations_success"] = float(self.iterations_success)
        return d


class GraphedTrainStep(_GraphOwner):
    """The whole training step as two HIP graphs around the (eager) RCCL all-reduces:

        graph A   preprocess, trunk, FPN, RPN, static-shape labelling/sampling, RoI heads, losses, backward
                  (gradients land in the flat gradient through the kernels' sinks / one multi-tensor add)
        eager     all-reduce of the loss vector and of the gradient buckets (world_size > 1 only)
        graph B   divergence guard, non-finite scan, fused SGD-momentum update, counters

    Possible because the dense training path has fixed shapes and no host<->device sync.  The reference clips the
    loss to [0,1] before backward when it diverges (train_net.py:212) but then discards that step's gradients
    (:259-261), so running backward on the unclipped local loss and skipping the update gives the same parameters.
    Per step the host only refreshes the static input buffers (image batch, padded ground truth, camera constants) and
    the schedule's learning-rate factor (a device scalar the captured update reads).  Replays are enqueued back to back
    (the host runs ahead of the device); CR_STEP_SYNC=1 makes the host wait for every step (debugging).

    Ownership: everything whose address is baked into the graphs is either a persistent buffer of this object / the
    optimizer (static inputs, flat parameter / gradient / momentum buffers, the weight bank, counters) or was allocated
    during capture from the graphs' private pool.  The ground-truth buffers hold `G` rows per image; a batch with more
    objects re-captures the graphs with a larger G (logged).
    """
    G_PAD = 32
    TOLERANCE, GAMMA = TrainStep.TOLERANCE, TrainStep.GAMMA

    def __init__(self, cfg, model, optimizer, sample_data, world_size=1, bucket_mb=32, warmup=2):
        from ..modeling.dense_train import GTBatch, camera_meta
        assert model.training and model.dense_train
        self.model, self.opt, self.world = model, optimizer, world_size
        self.stabilize = cfg.MODEL.STABILIZE > 0
        import os
        # Replays are enqueued back to back (the host runs ahead of the device); CR_STEP_SYNC=1 adds one host wait per step.
        # The captured region must not contain memset NODES: torch.topk's multi-block path zeroes its counters with
        # hipMemsetAsync, and with such nodes inside graph A back-to-back replays ended in GPU memory faults (5 of 5 runs of
        # `CR_GRAPHS=step python bench.py`, none with a host wait per step); with the own top-k (csrc/topk.hip, no memset)
        # 65 run-ahead replays per precision mode run clean.
        self.sync_each_step = os.environ.get("CR_STEP_SYNC", "0") == "1"
        self._GTBatch, self._camera_meta = GTBatch, camera_meta
        dev = optimizer.flat_p.device
        self.dev = dev
        self.warmup = warmup
        self.recent_loss = torch.full((), float("nan"), device=dev)
        self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self.iterations_success = torch.zeros((), device=dev)
        self.iterations_explode = torch.zeros((), device=dev)
        self.total = torch.zeros((), device=dev)
        n = optimizer.flat_g.numel()
        be = max(1, int(bucket_mb * (1 << 20) / 4))
        self.buckets = [(i, min(i + be, n)) for i in range(0, n, be)][::-1]
        self.comm_stream = torch.cuda.Stream(device=dev) if world_size > 1 else None
        model._graphed = None
        need = max(1, max(len(d["instances"]) for d in sample_data))
        self._capture(sample_data, max(self.G_PAD, need))

    def _capture(self, sample_data, G):
        """(re)build the static buffers for G ground-truth rows per image and capture both graphs"""
        from ..modeling.graphed import _fresh_leaves, capture_guard
        model, optimizer, dev, world_size = self.model, self.opt, self.dev, self.world
        optimizer.enable_weight_bank()
        self.dtype = ops.precision()
        self.G = int(G)
        images, batch = model._stack_images(sample_data)
        self.image_sizes = [tuple(s) for s in images.image_sizes]
        assert all(s == tuple(batch.shape[-2:]) for s in self.image_sizes), "whole-step graph: one image size per batch"
        self.static_img = batch.clone()
        self.gt = self._GTBatch([d["instances"].to(dev) for d in sample_data], dev, G=self.G)
        self.meta = self._meta_of(sample_data).clone()
        self.vals = None
        self.keys = None
        state = [t.clone() for t in (self.recent_loss, self.flag, self.iterations_success, self.iterations_explode)]

        def fwd_bwd():
            loss_dict = model.forward_static(self.static_img, self.image_sizes, self.gt, self.meta)
            keys = sorted(loss_dict.keys())
            vals = torch.stack([loss_dict[k].float() for k in keys])
            if self.vals is None:
                self.keys, self.vals = keys, torch.zeros_like(vals.detach())
                self.red = torch.zeros_like(self.vals)
            self.vals.copy_(vals.detach())
            optimizer.flat_g.zero_()
            leaves = [p for p in model.parameters() if p.requires_grad]
            grads = torch.autograd.grad(vals.sum(), leaves, allow_unused=True)
            views = [p._cr_grad for p, g in zip(leaves, grads) if g is not None]
            if views:
                torch._foreach_add_(views, [g for g in grads if g is not None])

        def update():
            ops.loss_guard(self.vals, 1.0 / world_size, self.red, self.total, self.recent_loss, self.stabilize, self.TOLERANCE,
                           self.GAMMA, self.flag)
            if self.stabilize:
                ops.nonfinite_flag(optimizer.flat_g, self.flag)
            optimizer.step(skip_flag=self.flag, grad_scale=1.0 / world_size)      # lr factor read from the device scalar
            ops.step_counters(self.flag, self.iterations_explode, self.iterations_success)

        # warm-up (eager, side stream), then restore every piece of state the warm-up touched
        snap = [t.clone() for t in (optimizer.flat_p, optimizer.flat_m)]
        bn = [(m, m.running_mean.clone(), m.running_var.clone(), m.num_batches_tracked.clone())
              for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        optimizer.refresh_lr()
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s), _fresh_leaves([model]):
            for _ in range(self.warmup):
                fwd_bwd()
                update()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)

        def restore():
            optimizer.flat_p.copy_(snap[0]); optimizer.flat_m.copy_(snap[1])
            for m, a, b, c in bn:
                m.running_mean.copy_(a); m.running_var.copy_(b); m.num_batches_tracked.copy_(c)
            for t, v in zip((self.recent_loss, self.flag, self.iterations_success, self.iterations_explode), state):
                t.copy_(v)
        restore()
        ops.bump_weight_epoch()
        with capture_guard() as self._keep, _fresh_leaves([model]):
            self.graph_a = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_a):
                fwd_bwd()
        self.graph_b = torch.cuda.CUDAGraph()
        with capture_guard(), torch.cuda.graph(self.graph_b, pool=self.graph_a.pool()):
            update()
        torch.cuda.synchronize(dev)
        restore()
        ops.bump_weight_epoch()
        self.last = {"keys": self.keys, "values": self.red, "total": self.total, "skipped": self.flag}

    def _meta_of(self, data):
        Ks = [torch.as_tensor(d["K"], dtype=torch.float32) for d in data]
        ratios = [d["height"] / s[0] for d, s in zip(data, self.image_sizes)]
        return self._camera_meta(self.model.roi_heads, Ks, ratios, self.image_sizes, self.dev)

    def load(self, data):
        """refresh the static inputs (device-side copies only; the one host->device transfer is pinned + async)"""
        need = max(1, max(len(d["instances"]) for d in data))
        if need > self.G:
            # more objects in an image than the captured ground-truth buffers hold: re-capture with room to spare
            # (the reference takes any number of boxes; crowded Omni3D images exceed 32)
            import logging
            G = 1 << (need - 1).bit_length()
            logging.getLogger(__name__).warning("GraphedTrainStep: %d ground-truth rows in an image > %d captured; "
                                                "re-capturing the step graphs with G = %d", need, self.G, G)
            torch.cuda.synchronize(self.dev)
            self._capture(data, G)
        _, batch = self.model._stack_images(data)
        assert tuple(batch.shape) == tuple(self.static_img.shape), "whole-step graph: the batch shape is fixed at capture"
        self.static_img.copy_(batch)
        self.gt.refill([d["instances"].to(self.dev) for d in data])
        meta = self._meta_of(data)
        self.meta.copy_(meta, non_blocking=True)

    def __call__(self, data):
        if ops.precision() != self.dtype:
            raise RuntimeError("GraphedTrainStep was captured in another precision mode (hipops.set_precision)")
        self.load(data)
        self.graph_a.replay()
        if self.world > 1:
            dist.all_reduce(self.vals)
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                for a, b in self.buckets:
                    dist.all_reduce(self.opt.flat_g[a:b])
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.opt.refresh_lr()                 # the schedule's factor for THIS update (graph B reads the device scalar)
        self.graph_b.replay()
        ops.bump_weight_epoch()               # parameters moved through raw pointers: cached compute copies are stale
        if self.sync_each_step:
            torch.cuda.current_stream(self.dev).synchronize()
        return self.last

    report = TrainStep.report
