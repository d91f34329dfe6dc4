"""HIP-graph capture of the static dense region of the train step: preprocess -> DLA trunk -> FPN -> RPN head
forward, and the matching backward.  Shapes there depend only on the image batch shape, so the ~700 kernel
launches (and their Python/ctypes issue cost) per direction collapse into one graph launch each
(MI355X_MICROARCH.md price list: eager goes host-bound below ~3 us per kernel).

The dynamic parts of the step (anchor/proposal sampling, RoI heads on a data-dependent number of boxes) stay
eager and talk to the graphs through static input / output / gradient buffers.
"""
import contextlib
import gc

import torch
import torch.nn as nn

from ... import hipops as ops


@contextlib.contextmanager
def capture_guard():
    """No cyclic garbage collection while a stream capture is open: a collection may run the destructor of an older
    CUDAGraph (graph-owning objects sit in reference cycles through their closures), and a graph / pool teardown inside
    a capture aborts the process on ROCm 7.2 (seen once as `Fatal Python error: Aborted ... Garbage-collecting`)."""
    was = gc.isenabled()
    gc.collect()
    _empty_graveyard()
    gc.disable()
    keep, prev = [], ops._PLAN_KEEP[0]
    ops._PLAN_KEEP[0] = keep       # objects whose buffers the captured kernels point into; the owner stores the list
    try:
        yield keep
    finally:
        ops._PLAN_KEEP[0] = prev
        if was:
            gc.enable()
        _empty_graveyard()


_GRAVEYARD = []      # state of graph owners whose destructor ran while a capture was open; emptied outside captures


def _bury_or_drain(owner):
    if owner.dev is None:
        return
    if torch.cuda.is_current_stream_capturing():
        # destroying a graph (or waiting for the device) inside an open capture invalidates it / aborts on ROCm 7.2: keep
        # the dead owner's graphs and buffers alive until the capture has closed (capture_guard empties the list)
        _GRAVEYARD.append(dict(owner.__dict__))
        return
    torch.cuda.synchronize(owner.dev)


def _empty_graveyard():
    if _GRAVEYARD and not torch.cuda.is_current_stream_capturing():
        torch.cuda.synchronize()
        _GRAVEYARD.clear()


class GraphOwner:
    """Base of the objects that own captured graphs: replays may still be in flight when the last reference goes away
    (run-ahead steps); the device is drained before the graphs are destroyed.  A destructor that runs while a stream
    capture is open (an explicit gc.collect(), a reference dropped by captured code) parks the graphs instead."""
    dev = None

    def __del__(self):
        try:
            _bury_or_drain(self)
        except Exception:
            pass


@contextlib.contextmanager
def _fresh_leaves(modules):
    """Temporarily replace every parameter of `modules` by a NEW nn.Parameter over the same storage.  Autograd
    identifies a leaf by its grad-accumulator node, which is pinned to the stream it was first used on; stale ones
    (from earlier eager steps) make the engine synchronise the capture stream with the default stream and the
    capture dies.  Fresh leaves get fresh accumulators on the capturing stream; memory addresses are unchanged,
    so the graphs keep reading / writing the optimizer's flat buffers."""
    swapped = []
    for mod in modules:
        for m in mod.modules():
            for name, p in list(m._parameters.items()):
                if p is None:
                    continue
                q = nn.Parameter(p.data, requires_grad=p.requires_grad)
                sink = ops.grad_sink(p)
                if sink is not None:
                    q._cr_grad = sink
                if hasattr(p, "_cr_bank"):
                    q._cr_bank = p._cr_bank
                m._parameters[name] = q
                swapped.append((m, name, p))
    try:
        yield
    finally:
        for m, name, p in swapped:
            m._parameters[name] = p


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, trigger):
        runner.fwd_graph.replay()
        ctx.runner = runner
        ctx.set_materialize_grads(False)       # an output nobody differentiates arrives as None (one fill below, not fill + copy)
        return tuple(o.detach() for o in runner.static_outs)

    @staticmethod
    def backward(ctx, *grads):
        r = ctx.runner
        if r.pre_bwd is not None:          # every gradient of the RoI heads is final here (they feed this node's inputs)
            r.pre_bwd()
        for sg, g in zip(r.static_grads, grads):
            if g is None:
                sg.zero_()
            elif g.data_ptr() != sg.data_ptr():      # RoIAlign scatters straight into the static buffer (_cr_grad_dst)
                sg.copy_(g)
        r.bwd_graph.replay()
        if r.bwd_graph2 is not None:       # two-segment backward: level5 + FPN + RPN head | the rest of the trunk
            if r.mid_bwd is not None:      # the first segment's gradients are final: their all-reduce overlaps the second
                r.mid_bwd()
            r.bwd_graph2.replay()
        return None, None


class GraphedDense(GraphOwner):
    """split_backward=True (data-parallel runs) captures the backward as TWO graphs cut at the input of the trunk's last
    level: [RPN head, FPN, DLA level5] first, [level4 ... stem] second.  The first segment owns 2/3 of the region's
    parameters and a small part of its backward time, so their gradient all-reduce (TrainStep, `mid_bwd`) runs on the
    communication stream under the second segment."""

    def __init__(self, model, images_u8, warmup=2, split_backward=False):
        assert model.training, "capture the training-mode dense region"
        self.model = model
        self.shape = tuple(images_u8.shape)
        self.dtype = ops.precision()        # the captured kernels are those of this precision mode
        dev = self.dev = images_u8.device
        self.static_img = images_u8.clone()
        self.pre_bwd = None                 # optional callback run right before the backward graph is replayed
        self.mid_bwd = None                 # optional callback run between the two backward segments
        self.bwd_graph2 = None
        bu = getattr(model.backbone, "bottom_up", None)
        lower = [getattr(bu, n, None) for n in ("base_layer", "level0", "level1", "level2", "level3", "level4")]
        self.split = bool(split_backward) and bu is not None and all(m is not None for m in lower) and hasattr(bu, "level5")
        cut = {}
        self.trigger = torch.zeros((), device=dev, requires_grad=True)
        pg = model.proposal_generator
        self.feat_names = None

        def dense():
            x = ops.preprocess(self.static_img, model.pixel_mean_list, model.pixel_std_list)
            feats = model.backbone(x)
            ys = pg.rpn_head.forward_raw([feats[f] for f in pg.in_features])     # (B,H,W,16) per level: unpacked outside
            self.feat_names = list(feats.keys())
            self.n_levels = len(ys)
            return tuple(feats.values()) + tuple(ys)

        mods = [model.backbone, pg.rpn_head]

        def leaves():
            ps = [p for m in mods for p in m.parameters() if p.requires_grad]
            for p in ps:
                assert ops.grad_sink(p) is not None, "build the optimizer (FlatSGD) before capturing graphs"
            return ps

        def lower_ids():
            return {id(p) for m in lower for p in m.parameters()}

        def segment_params(first):
            """parameters (of the live modules) whose gradients the first / second backward segment produces"""
            low = lower_ids()
            return [p for m in mods for p in m.parameters() if p.requires_grad and ((id(p) not in low) == first)]
        self.segment_params = segment_params

        def run_backward(outs, grads):
            # torch.autograd.grad, not .backward(): no AccumulateGrad nodes (they are pinned to the stream they were
            # first created on, which breaks capture).  Conv / BN kernels accumulate into the flat gradient
            # themselves and return None; what comes back here went through plain autograd (biases, the stem and
            # predictor weights) and is added to the flat gradient inside the captured region.
            ps = leaves()
            # an output that a convolution of the region consumed first (the pyramid maps -> RPN head) has a gradient slot:
            # the incoming gradient goes there and that convolution's backward-data adds it in its epilogue
            keep_o, keep_g = [], []
            for o, g in zip(outs, grads):
                slot = getattr(o, "_cr_slot", None)
                if slot is not None and ops._SLOTS_ON[0]:
                    ops._slot_put(slot, g)
                else:
                    keep_o.append(o); keep_g.append(g)
            if not self.split:
                res = torch.autograd.grad(keep_o, ps, keep_g, allow_unused=True)
                for p, g in zip(ps, res):
                    if g is not None:
                        ops.grad_sink(p).add_(g)
                return None
            # first segment: everything downstream of the cut tensor (level4's output: consumed by level5 and the FPN
            # lateral, both in this segment; the laterals of the lower levels leave their contributions in the gradient
            # slots of level2 / level3's outputs, where the second segment's convolutions pick them up)
            low = lower_ids()
            ps1 = [p for p in ps if id(p) not in low]
            res = torch.autograd.grad(keep_o, ps1 + [cut["x"]], keep_g, allow_unused=True, retain_graph=True)
            for p, g in zip(ps1, res[:-1]):
                if g is not None:
                    ops.grad_sink(p).add_(g)
            assert res[-1] is not None, "two-segment backward: no gradient reaches the trunk"
            return res[-1]

        def run_backward2(gx):
            low = lower_ids()
            ps2 = [p for p in leaves() if id(p) in low]
            res = torch.autograd.grad([cut["x"]], ps2, [gx], allow_unused=True)
            for p, g in zip(ps2, res):
                if g is not None:
                    ops.grad_sink(p).add_(g)

        # the warm-up and capture passes below run the BatchNorm layers in training mode: their running statistics are
        # put back afterwards, so capturing a shape in the middle of a run (RCNN3D._train_graph_for) leaves no trace
        bn_state = [(m, m.running_mean.clone(), m.running_var.clone(), m.num_batches_tracked.clone())
                    for mod in mods for m in mod.modules()
                    if isinstance(m, nn.BatchNorm2d) and m.running_mean is not None]
        # the weight bank whose buffers the captured kernels read (FlatSGD keeps one bank per precision mode alive)
        self.bank = next((getattr(p, "_cr_bank")[0] for mod in mods for p in mod.parameters() if hasattr(p, "_cr_bank")), None)
        # ---- eager warm-up on a side stream (allocator / lazy-init effects out of the capture)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        hook = bu.level4.register_forward_hook(lambda m, i, o: cut.__setitem__("x", o)) if self.split else None
        with torch.cuda.stream(s), _fresh_leaves(mods):
            for _ in range(warmup):
                outs = dense()
                gx = run_backward(outs, tuple(torch.zeros_like(o) for o in outs))
                if self.split:
                    run_backward2(gx)
            del outs
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)

        # ---- capture.  The bf16 weight copies must be re-made INSIDE the graphs on every replay.
        ops.bump_weight_epoch()
        with capture_guard() as self._keep, _fresh_leaves(mods):
            self.fwd_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.fwd_graph):
                self.static_outs = dense()
            self.static_grads = tuple(torch.zeros_like(o) for o in self.static_outs)
            self.bwd_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.bwd_graph, pool=self.fwd_graph.pool()):
                gx = run_backward(self.static_outs, self.static_grads)
            if self.split:
                self.bwd_graph2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.bwd_graph2, pool=self.fwd_graph.pool()):
                    run_backward2(gx)
        if hook is not None:
            hook.remove()
        cut.clear()
        torch.cuda.synchronize(dev)
        for m, a, b, c in bn_state:
            m.running_mean.copy_(a); m.running_var.copy_(b); m.num_batches_tracked.copy_(c)
        self._bank_param = next((p for mod in mods for p in mod.parameters() if hasattr(p, "_cr_bank")), None)

    def matches(self, images_u8):
        """same batch shape, device and precision mode, and the weight bank the graphs were captured against is still the
        one attached to the parameters (a bank rebuilt by the optimizer means new compute-copy buffers)"""
        bp = self._bank_param
        bank_now = getattr(bp, "_cr_bank", (None,))[0] if bp is not None else None
        return (tuple(images_u8.shape) == self.shape and images_u8.device == self.static_img.device
                and ops.precision() == self.dtype and bank_now is self.bank)

    def __call__(self, images_u8):
        self.static_img.copy_(images_u8)
        outs = _Replay.apply(self, self.trigger)
        nf, nl = len(self.feat_names), self.n_levels
        for o, sg in zip(outs[:nf], self.static_grads):
            o._cr_grad_dst = sg            # hipops._ROIAlign: scatter the pyramid gradient straight into the static buffer
        feats = dict(zip(self.feat_names, outs[:nf]))
        return feats, list(outs[nf:nf + nl])


class GraphedDenseEval(GraphOwner):
    """Forward-only HIP graph of the static dense region in eval mode (preprocess, trunk with BatchNorm on running
    statistics, FPN, RPN head) for one image-batch shape: inference is launch-bound at 8 images per step."""

    def __init__(self, model, images_u8, warmup=2):
        assert not model.training, "capture the eval-mode dense region"
        self.shape = tuple(images_u8.shape)
        self.dtype = ops.precision()
        dev = self.dev = images_u8.device
        self.static_img = images_u8.clone()
        pg = model.proposal_generator

        def dense():
            x = ops.preprocess(self.static_img, model.pixel_mean_list, model.pixel_std_list)
            feats = model.backbone(x)
            logits, deltas = pg.rpn_head([feats[f] for f in pg.in_features])
            self.feat_names = list(feats.keys())
            self.n_levels = len(logits)
            return tuple(feats.values()) + tuple(logits) + tuple(deltas)

        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(warmup):
                dense()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        ops.bump_weight_epoch()          # the bf16 weight copies are made inside the graph (cheap; weights may change)
        self.graph = torch.cuda.CUDAGraph()
        # the BatchNorm folds are NOT part of the graph: it reads persistent folded weights / biases that refresh_folds()
        # rewrites before a replay when a parameter, a running statistic or the weight epoch has moved (serving: never)
        self.folds = []
        ops._FOLD_REG[0] = self.folds
        try:
            with capture_guard() as self._keep, torch.no_grad(), torch.cuda.graph(self.graph):
                self.static_outs = dense()
        finally:
            ops._FOLD_REG[0] = None
        torch.cuda.synchronize(dev)
        ops.refresh_folds(self.folds)

    def matches(self, images_u8):
        return (tuple(images_u8.shape) == self.shape and images_u8.device == self.static_img.device
                and ops.precision() == self.dtype)

    def __call__(self, images_u8):
        self.static_img.copy_(images_u8)
        ops.refresh_folds(self.folds)
        self.graph.replay()
        outs = self.static_outs
        nf, nl = len(self.feat_names), self.n_levels
        return dict(zip(self.feat_names, outs[:nf])), list(outs[nf:nf + nl]), list(outs[nf + nl:nf + 2 * nl])
