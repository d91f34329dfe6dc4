"""HIP-graph capture of the static dense region of the train step: preprocess -> DLA trunk -> FPN -> RPN head
forward, and the matching backward.  Shapes there depend only on the image batch shape, so the ~700 kernel
launches (and their Python/ctypes issue cost) per direction collapse into one graph launch each
(MI355X_MICROARCH.md price list: eager goes host-bound below ~3 us per kernel).

The dynamic parts of the step (anchor/proposal sampling, RoI heads on a data-dependent number of boxes) stay
eager and talk to the graphs through static input / output / gradient buffers.
"""
import torch

from ... import hipops as ops


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, trigger):
        runner.fwd_graph.replay()
        ctx.runner = runner
        return tuple(o.detach() for o in runner.static_outs)

    @staticmethod
    def backward(ctx, *grads):
        r = ctx.runner
        for sg, g in zip(r.static_grads, grads):
            if g is None:
                sg.zero_()
            else:
                sg.copy_(g)
        r.bwd_graph.replay()
        return None, None


class GraphedDense:
    def __init__(self, model, images_u8, warmup=2):
        assert model.training, "capture the training-mode dense region"
        self.model = model
        self.shape = tuple(images_u8.shape)
        dev = images_u8.device
        self.static_img = images_u8.clone()
        self.trigger = torch.zeros((), device=dev, requires_grad=True)
        pg = model.proposal_generator
        self.feat_names = None

        def dense():
            x = ops.preprocess(self.static_img, model.pixel_mean_list, model.pixel_std_list)
            feats = model.backbone(x)
            logits, deltas = pg.rpn_head([feats[f] for f in pg.in_features])
            self.feat_names = list(feats.keys())
            self.n_levels = len(logits)
            return tuple(feats.values()) + tuple(logits) + tuple(deltas)

        params = [p for p in model.backbone.parameters()] + [p for p in pg.rpn_head.parameters()]
        self.extra = [p for p in params if p.requires_grad]
        for p in self.extra:
            assert ops.grad_sink(p) is not None, "build the optimizer (FlatSGD) before capturing graphs"

        def run_backward(outs, grads):
            for p in self.extra:
                p.grad = None
            torch.autograd.backward(outs, grads)
            # gradients that came through plain autograd (biases, stem / predictor weights) -> flat gradient
            for p in self.extra:
                if p.grad is not None:
                    ops.grad_sink(p).add_(p.grad)

        # ---- eager warm-up on a side stream (allocator / lazy-init effects out of the capture)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                outs = dense()
                run_backward(outs, tuple(torch.zeros_like(o) for o in outs))
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)

        # ---- capture.  The bf16 weight copies must be re-made INSIDE the graphs on every replay.
        ops.bump_weight_epoch()
        self.fwd_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd_graph):
            self.static_outs = dense()
        self.static_grads = tuple(torch.zeros_like(o) for o in self.static_outs)
        self.bwd_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.bwd_graph, pool=self.fwd_graph.pool()):
            run_backward(self.static_outs, self.static_grads)
        self._static_param_grads = [p.grad for p in self.extra]       # keep the captured buffers alive
        for p in self.extra:
            p.grad = None
        torch.cuda.synchronize(dev)

    def matches(self, images_u8):
        return tuple(images_u8.shape) == self.shape and images_u8.device == self.static_img.device

    def __call__(self, images_u8):
        self.static_img.copy_(images_u8)
        outs = _Replay.apply(self, self.trigger)
        nf, nl = len(self.feat_names), self.n_levels
        feats = dict(zip(self.feat_names, outs[:nf]))
        return feats, list(outs[nf:nf + nl]), list(outs[nf + nl:nf + 2 * nl])
