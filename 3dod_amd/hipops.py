"""torch.autograd glue over the C-ABI conv / BN / pooling / ROIAlign / NMS / optimizer kernels.

PyTorch is used for device memory, streams and the autograd tape only; every
op below is one or a few calls into libcr3dod.so.  Activations are NHWC
tensors in the ACTIVATION DTYPE of the process; conv weights are float32
(Cout,Cin,k,k) parameters in channels_last memory format (physical
[Cout][k*k][Cin]).  There is no CPU path: a non-CUDA tensor raises CrError.

Precision.  The reference trains and evaluates in float32 (tools/train_net.py:184-330,
no autocast anywhere), so float32 is the default here: activations, activation
gradients and the operands of every contraction are f32 and the convolutions / FC
layers run on the f32 MFMA (v_mfma_f32_16x16x4_f32).  `set_precision("bf16")` (or
CR_PRECISION=bf16 in the environment) selects the fast mode: the same tensors are
bfloat16 and the contractions run on the bf16 MFMA with f32 accumulation.  Every op
takes its mode from the dtype of the tensor it is handed, so both modes can coexist
in one process (the parity tests run them side by side).
"""
import ctypes
import os

import torch

from . import _lib

bf16 = torch.bfloat16
f32 = torch.float32
STAT_REPL = 32

# name -> (activation dtype, act_f32 flag of the C ABI)
_PRECISIONS = {"fp32": (f32, 1), "f32": (f32, 1), "float32": (f32, 1), "fp32x3": (f32, 2), "bf16": (bf16, 0),
               "bfloat16": (bf16, 0)}
_ACT = [_PRECISIONS[os.environ.get("CR_PRECISION", "fp32").lower()]]


def set_precision(name):
    """"fp32" (reference precision on the f32 MFMA, default), "fp32x3" (float32 storage and float32-accurate
    contractions on the bf16 matrix cores: every operand split exactly into three bf16 values, six products per term --
    include/cr3dod.h, act_f32 = 2) or "bf16" (fast mode).  Decides the dtype of the activations produced by preprocess()
    and hence of everything downstream, and which kernels f32 tensors run on.  Returns the previous setting's name."""
    prev = precision()
    _ACT[0] = _PRECISIONS[str(name).lower()]
    return prev


def precision():
    return {0: "bf16", 1: "fp32", 2: "fp32x3"}[_ACT[0][1]]


def act_dtype():
    return _ACT[0][0]


def _af(t):
    """act_f32 flag of the C ABI from a tensor's dtype (f32 tensors: 1, or 2 in the split mode)"""
    if t.dtype == f32:
        return _ACT[0][1] if _ACT[0][0] == f32 else 1
    if t.dtype == bf16:
        return 0
    raise _lib.CrError(f"activations must be float32 or bfloat16, got {t.dtype}")


def _w3(w, af):
    """split-mode companion (cr_weight_split3) of a prepared f32 weight matrix (rows, K), or None.  WeightBank views
    carry theirs (refreshed with the bank); anything else is split on first use per weight epoch and cached on the tensor."""
    if af != 2 or w.dtype != f32:
        return None
    ent = getattr(w, "_cr_w3", None)
    if ent is not None and ent[0] == "bank":
        return ent[1]
    rows = w.shape[0]                      # (rows, K) or a channels_last (Cout,Cin,k,k) weight = physical [Cout][k*k*Cin]
    K = w.numel() // rows
    if K % 32:
        return None
    tag = (_WEIGHT_EPOCH[0], w.data_ptr(), w._version)
    if ent is None or ent[0] != tag:
        _p = _Args()
        out = torch.empty((rows * K * 3,), dtype=bf16, device=w.device)
        _chk(_lib.load().cr_weight_split3(_ctx(w), _p(w), _p(out), rows, K), "cr_weight_split3")
        ent = (tag, out)
        try:
            w._cr_w3 = ent
        except Exception:
            pass
    return ent[1]


def _ctx(t):
    return _lib.ctx_for(t.device)


def _chk(rc, what):
    _lib.check(rc, what)


class _Args:
    """Owns the temporaries of ONE kernel call.  Every wrapper below starts with `_p = _Args()` and takes its device
    pointers through it: `_p(x.contiguous())` may create a temporary, which this object keeps referenced until the wrapper
    returns -- i.e. until the launch has been enqueued on the stream.  After that the caching allocator's stream ordering
    (eager) or the capture's allocation order (HIP graphs) makes reuse of the block safe.  Without an owner a temporary dies
    as soon as its pointer has been read and the NEXT temporary of the same argument list can be handed the same block
    before the kernel is even launched (round 1: garbage sampling indices -> out-of-bounds gathers in cr_roi_compact, the
    memory fault of the whole-step graph mode)."""
    __slots__ = ("keep",)

    def __init__(self):
        self.keep = []

    def __call__(self, t):
        if t is not None:
            self.keep.append(t)
        return _lib.ptr(t)


def _p(t):
    """device pointer of a tensor the CALLER keeps alive across the launch (NULL for None)"""
    return _lib.ptr(t)


def _need_cuda(t, name):
    if not t.is_cuda:
        raise _lib.CrError(f"{name}: expected a CUDA(HIP) tensor; 3dod_amd has no CPU path")


# --------------------------------------------------------------------------
# weight preparation (cached per parameter version)
# --------------------------------------------------------------------------
_WEIGHT_EPOCH = [0]


def bump_weight_epoch():
    """call after parameters were updated outside torch's version tracking (cr_sgd_step writes through raw
    pointers): invalidates every cached bf16 weight copy."""
    _WEIGHT_EPOCH[0] += 1


def as_krsc(weight):
    """(Cout,Cin,k,k) f32 parameter whose storage is channels_last = physical [Cout][k*k*Cin]."""
    if weight.dim() != 4:
        raise ValueError("conv weight must be 4-D")
    if not weight.is_contiguous(memory_format=torch.channels_last):
        raise _lib.CrError("conv weight must be channels_last (see cubercnn.modeling.backbone.to_channels_last)")
    return weight


class WeightBank:
    """bf16 compute copies ([Cout][k*k*Cin] and the bwd-data layout [Cin][k*k*Cout]) of ALL registered conv weights,
    refreshed by ONE launch (cr_weights_prepare) the first time any of them is asked for after the weight epoch
    moved (= after an optimizer step).  The weights must be views into one flat f32 buffer (FlatSGD.flat_p).
    Activated by the training step objects; anything that changes weights outside the optimizer must call
    bump_weight_epoch() (the model's load_state_dict hook does)."""

    def __init__(self, params, flat_p, dtype=None, split=None):
        import numpy as np
        dev = flat_p.device
        self.flat_p = flat_p
        self.dtype = dtype = dtype if dtype is not None else act_dtype()
        self.split = split = (dtype == f32 and _ACT[0][1] == 2) if split is None else bool(split and dtype == f32)
        self.mode = "bf16" if dtype == bf16 else ("fp32x3" if split else "fp32")
        descs, tiles, self.shapes = [], [], []
        off = 0
        for i, p in enumerate(params):
            Cout, Cin, k, _ = p.shape
            KK, n = k * k, p.numel()
            src_off = (p.data_ptr() - flat_p.data_ptr()) // 4
            assert 0 <= src_off and src_off + n <= flat_p.numel(), "conv weight is not a view of the flat buffer"
            descs.append((src_off, off, off, Cout, KK, Cin, 1))
            for r in range(KK):
                for o0 in range(0, Cout, 32):
                    for c0 in range(0, Cin, 32):
                        tiles.append((i, r, o0, c0))
            self.shapes.append((off, n, Cout, KK, Cin))
            off += (n + 7) // 8 * 8
        dt = np.dtype([("src", "<i8"), ("dst", "<i8"), ("dstT", "<i8"), ("Cout", "<i4"), ("KK", "<i4"), ("Cin", "<i4"),
                       ("needT", "<i4")])
        assert dt.itemsize == 40
        self.descs = torch.from_numpy(np.array(descs, dtype=dt).view(np.uint8).copy()).to(dev)
        self.tiles = torch.tensor(tiles, dtype=torch.int32, device=dev)
        self.ntiles = len(tiles)
        # f32 mode: the master weights ARE the forward / weight-gradient operand; only the bwd-data layout is a copy
        self.dst = torch.empty((off,), dtype=bf16, device=dev) if dtype == bf16 else None
        self.dstT = torch.empty((off,), dtype=dtype, device=dev)
        if dtype == bf16:
            self.views = [(self.dst[o:o + n].view(Cout, KK * Cin), self.dstT[o:o + n].view(Cin, KK * Cout))
                          for (o, n, Cout, KK, Cin) in self.shapes]
        else:
            self.views = [(p.detach(), self.dstT[o:o + n].view(Cin, KK * Cout))
                          for p, (o, n, Cout, KK, Cin) in zip(params, self.shapes)]
        if split:
            # split-mode planes (6 bytes per element) of every weight whose k extent is a multiple of 32, forward layout
            # from the flat parameter buffer and backward-data layout from dstT, one launch each
            s3 = np.dtype([("src", "<i8"), ("dst", "<i8"), ("item0", "<i8"), ("rows", "<i4"), ("K", "<i4")])
            assert s3.itemsize == 32
            recs = ([], [])
            tot, o3 = [0, 0], 0
            for i, (p, (o, n, Cout, KK, Cin)) in enumerate(zip(params, self.shapes)):
                src_off = (p.data_ptr() - flat_p.data_ptr()) // 4
                for which, (soff, rows, K) in enumerate(((src_off, Cout, KK * Cin), (o, Cin, KK * Cout))):
                    if K % 32 or soff % 4:
                        continue
                    recs[which].append((soff, o3, tot[which], rows, K, i))
                    tot[which] += rows * K // 8
                    o3 += rows * K * 3
            self.w3 = torch.empty((max(o3, 8),), dtype=bf16, device=dev)
            self.s3 = []
            for which in (0, 1):
                arr = np.array([r[:5] for r in recs[which]], dtype=s3) if recs[which] else np.zeros((0,), dtype=s3)
                self.s3.append((torch.from_numpy(arr.view(np.uint8).copy()).to(dev), len(recs[which]), tot[which]))
                for (soff, d3, _, rows, K, i) in recs[which]:
                    self.views[i][which]._cr_w3 = ("bank", self.w3[d3:d3 + rows * K * 3])
        self.epoch = None
        self.params = list(params)
        self.attach()

    def attach(self):
        """make this bank the one the parameters' compute copies come from (FlatSGD keeps one bank per precision mode)"""
        for i, p in enumerate(self.params):
            p._cr_bank = (self, i)

    def get(self, i):
        _p = _Args()
        if self.epoch != _WEIGHT_EPOCH[0]:
            lib = _lib.load()
            _chk(lib.cr_weights_prepare(_ctx(self.flat_p), _p(self.flat_p), _p(self.dst), _p(self.dstT), _p(self.descs),
                                        _p(self.tiles), self.ntiles, int(self.dtype == f32)), "cr_weights_prepare")
            if self.split:
                for src, (descs, nd, tot) in zip((self.flat_p, self.dstT), self.s3):
                    _chk(lib.cr_weights_split3(_ctx(self.flat_p), _p(src), _p(self.w3), _p(descs), nd, tot), "cr_weights_split3")
            self.epoch = _WEIGHT_EPOCH[0]
        return self.views[i]


def prepared_weights(weight, need_transposed, dtype=bf16):
    """compute copies of a f32 channels_last weight in the activation dtype: [Cout][k*k*Cin] and (optionally) the bwd-data
    layout [Cin][k*k*Cout].  In f32 mode the first one is the weight itself.  Weights registered in a WeightBank of the
    same dtype come from the bank (one launch per step for the whole model); others are cached ON the tensor object (so
    a new tensor at a recycled address never hits a stale entry), keyed by torch's version counter and the global
    weight epoch."""
    _p = _Args()
    bk = getattr(weight, "_cr_bank", None)
    if bk is not None and bk[0].dtype == dtype and (dtype == bf16 or bk[0].split or _ACT[0][1] != 2):
        return bk[0].get(bk[1])
    attr = "_cr_wcache" if dtype == bf16 else "_cr_wcache32"
    ent = getattr(weight, attr, None)
    tag = (weight._version, _WEIGHT_EPOCH[0], weight.data_ptr())
    lib = _lib.load()
    if ent is None or ent[0] != tag:
        Cout, Cin, k, _ = weight.shape
        if dtype == bf16:
            wb = torch.empty((Cout, k * k * Cin), dtype=bf16, device=weight.device)
            wd = weight.detach()
            _chk(lib.cr_cast_f32_to_bf16(_ctx(weight), _p(wd), _p(wb), weight.numel()), "cr_cast_f32_to_bf16")
        else:
            wb = weight.detach()
        ent = [tag, wb, None]
        try:
            setattr(weight, attr, ent)
        except Exception:
            pass
    if need_transposed and ent[2] is None:
        Cout, Cin, k, _ = weight.shape
        wt = torch.empty((Cin, k * k * Cout), dtype=dtype, device=weight.device)
        wd = weight.detach()
        _chk(lib.cr_weight_transpose(_ctx(weight), _p(wd), _p(wt), Cout, k, Cin, int(dtype == f32)), "cr_weight_transpose")
        ent[2] = wt
    return ent[1], ent[2]


# --------------------------------------------------------------------------
# raw kernels
# --------------------------------------------------------------------------
def conv_fwd_raw(x, wb, Cout, k, stride, pad, bias=None, residual=None, relu=False, stats=None, out_f32=False):
    _p = _Args()
    _need_cuda(x, "conv input")
    af = _af(x)
    assert x.is_contiguous() and x.dim() == 4 and wb.dtype == x.dtype and (residual is None or residual.dtype == x.dtype)
    N, H, W, Cin = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty((N, Ho, Wo, Cout), dtype=f32 if (out_f32 or af) else bf16, device=x.device)
    lib = _lib.load()
    _chk(lib.cr_conv2d_fwd(_ctx(x), _p(x), _p(wb), _p(y), N, H, W, Cin, Cout, k, stride, pad, _p(bias), _p(residual),
                           int(relu), _p(stats), int(out_f32), af, _p(_w3(wb, af))), "cr_conv2d_fwd")
    return y


def conv_bwd_data_raw(dy, wt, in_shape, k, stride, pad, accumulate=None):
    """accumulate: a tensor of the result's shape that is added in the kernel's epilogue (see _GradSlot)"""
    _p = _Args()
    N, H, W, Cin = in_shape
    Cout = dy.shape[3]
    assert wt.dtype == dy.dtype
    dx = torch.empty((N, H, W, Cin), dtype=dy.dtype, device=dy.device)
    if accumulate is not None:
        assert tuple(accumulate.shape) == (N, H, W, Cin), "accumulate must have the input's shape"
        accumulate = accumulate.to(dy.dtype).contiguous()
    lib = _lib.load()
    af = _af(dy)
    _chk(lib.cr_conv2d_bwd_data(_ctx(dy), _p(dy), _p(wt), _p(dx), N, H, W, Cin, Cout, k, stride, pad, af, _p(_w3(wt, af)),
                                _p(accumulate)), "cr_conv2d_bwd_data")
    return dx


# --------------------------------------------------------------------------
# Fan-in of gradients without add kernels.  An activation consumed by several of the ops below would have its gradient
# contributions summed by autograd with one elementwise add per extra consumer (60 launches per train step).  Instead the
# consumers of a tensor share a _GradSlot: the FIRST consumer registered in the forward pass must be a convolution -- its
# backward runs last (autograd executes in reverse creation order, and where the later consumers feed the first one's
# output, as in a residual block, also by dependency) and adds what the others left in the slot inside its backward-data
# epilogue (`accumulate`).  The other consumers put their contribution into the slot and return None.  The order is
# CHECKED: a contribution arriving after the first consumer's backward ran raises (a registered consumer that receives no
# gradient at all simply never contributes).  Consumers that are plain torch ops do not take part (autograd adds their share
# as before).
# --------------------------------------------------------------------------
class _GradSlot:
    __slots__ = ("n_reg", "n_arr", "buf", "done", "what")

    def __init__(self, what=""):
        self.n_reg, self.n_arr, self.buf, self.done, self.what = 0, 0, None, False, what


_SLOTS_ON = [os.environ.get("CR_GRAD_SLOTS", "1") == "1"]


def _slot_register(x, accumulator):
    """called in the forward pass by a consumer of x; -> (slot, index) or (None, 0).  accumulator: this consumer can add the
    slot's content in its backward (a convolution's backward-data)"""
    if not (_SLOTS_ON[0] and torch.is_grad_enabled() and torch.is_tensor(x) and x.requires_grad):
        return None, 0
    slot = getattr(x, "_cr_slot", None)
    if slot is None:
        if not accumulator:
            return None, 0                    # the first consumer cannot accumulate: everybody returns gradients as usual
        slot = _GradSlot(f"tensor {tuple(x.shape)}")
        try:
            x._cr_slot = slot
        except Exception:
            return None, 0
    slot.n_reg += 1
    return slot, slot.n_reg


def _slot_put(slot, g):
    """a later consumer leaves its gradient contribution (backward pass); the caller returns None for that input"""
    if slot.done:
        raise RuntimeError("gradient slot: a contribution arrived after the accumulating convolution's backward had run "
                           "(set CR_GRAD_SLOTS=0 to fall back to autograd's adds)")
    slot.buf = g if slot.buf is None else slot.buf + g
    slot.n_arr += 1


def _slot_fold(slot):
    """a later consumer whose own kernel can add: takes what the slot holds so far (or None) and will put back the sum, so
    that several contributions to one slot never meet in a torch add"""
    buf, slot.buf = slot.buf, None
    return buf


def _slot_take(slot):
    """the first consumer (a convolution) collects what the others have contributed, in its backward.  A registered
    consumer whose output receives no gradient never contributes (e.g. a pooled level nobody uses): that is a zero, not an
    error; one that contributes AFTER this point raises in _slot_put -- nothing is ever dropped silently."""
    slot.done = True
    buf, slot.buf = slot.buf, None
    return buf


def grad_sink(t):
    """a persistent f32 gradient buffer attached to a parameter by the optimizer (FlatSGD): kernels accumulate
    straight into it (no per-parameter zero-fill / add kernels); the autograd grad for that input is None."""
    return getattr(t, "_cr_grad", None)


def conv_bwd_weight_raw(dy, x, k, stride, pad, sink=None, bias_acc=None):
    """dW (into `sink` when given).  bias_acc: f32 [Cout] buffer that additionally receives += sum_pixels dy (fused)."""
    _p = _Args()
    N, H, W, Cin = x.shape
    Cout = dy.shape[3]
    lib = _lib.load()
    af = _af(x)
    assert dy.dtype == x.dtype
    if bias_acc is not None:
        dw = sink if sink is not None else torch.empty((Cout, Cin, k, k), dtype=f32, device=x.device).contiguous(
            memory_format=torch.channels_last)
        _chk(lib.cr_conv2d_bwd_weight_bias(_ctx(x), _p(dy), _p(x), _p(dw), _p(bias_acc), N, H, W, Cin, Cout, k, stride, pad,
                                           int(sink is not None), af), "cr_conv2d_bwd_weight_bias")
        return None if sink is not None else dw
    if sink is not None:
        _chk(lib.cr_conv2d_bwd_weight(_ctx(x), _p(dy), _p(x), _p(sink), N, H, W, Cin, Cout, k, stride, pad, 1, af),
             "cr_conv2d_bwd_weight")
        return None
    dw = torch.empty((Cout, Cin, k, k), dtype=f32, device=x.device).contiguous(memory_format=torch.channels_last)
    _chk(lib.cr_conv2d_bwd_weight(_ctx(x), _p(dy), _p(x), _p(dw), N, H, W, Cin, Cout, k, stride, pad, 0, af),
         "cr_conv2d_bwd_weight")
    return dw


# --------------------------------------------------------------------------
# conv + BatchNorm(train) + residual + ReLU   (dla.py BasicBlock / Root / conv levels)
# --------------------------------------------------------------------------
class _ConvBN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, running_mean, running_var, stride, pad, relu, eps, momentum,
                training, slots=((None, 0), (None, 0))):
        ctx.slots = slots
        _p = _Args()
        Cout, Cin, k, _ = weight.shape
        need_grad = x.requires_grad or weight.requires_grad
        wb, wt = prepared_weights(weight, need_grad and x.requires_grad, x.dtype)
        dev = x.device
        af = _af(x)
        lib = _lib.load()
        if training:
            _STATS_EPOCH[0] += 1
            N_, H_, W_, _ = x.shape
            M = N_ * ((H_ + 2 * pad - k) // stride + 1) * ((W_ + 2 * pad - k) // stride + 1)
            nparts = (M + 63) // 64            # statistics rows are per 64 pixels (independent of the tile choice)
            stats = torch.empty((nparts, 2, Cout), dtype=f32, device=dev)
            y_raw = conv_fwd_raw(x, wb, Cout, k, stride, pad, stats=stats)
            out = torch.empty_like(y_raw)
            mi = torch.empty((2, Cout), dtype=f32, device=dev)
            _chk(lib.cr_bn_fwd(_ctx(x), _p(y_raw), _p(stats), nparts, _p(gamma.detach()), _p(beta.detach()), _p(residual),
                               _p(out), M, Cout, int(relu), float(eps), float(momentum), _p(mi), _p(running_mean),
                               _p(running_var), af), "cr_bn_fwd")
        else:
            # frozen statistics with gradients enabled (freeze_bn fine-tuning; plain inference takes conv_bn_folded): an
            # affine epilogue (scale*x + shift) done by the BN kernel with mean/invstd taken from the running buffers
            y_raw = conv_fwd_raw(x, wb, Cout, k, stride, pad)
            M = y_raw.numel() // Cout
            mi = torch.stack([running_mean, torch.rsqrt(running_var + eps)]).contiguous()
            out = torch.empty_like(y_raw)
            # one "partial" that reproduces (mean, var): sum = mean*M, sumsq = (var+mean^2)*M
            zstats = torch.stack([running_mean * M, (running_var + running_mean * running_mean) * M]).view(1, 2, Cout).contiguous()
            _chk(lib.cr_bn_fwd(_ctx(x), _p(y_raw), _p(zstats), 1, _p(gamma.detach()), _p(beta.detach()), _p(residual),
                               _p(out), M, Cout, int(relu), float(eps), 0.0, _p(mi), _p(None), _p(None), af), "cr_bn_fwd")
        ctx.cfg = (k, stride, pad, relu, training, residual is not None)
        ctx.beta_ref = beta
        ctx.save_for_backward(x, weight, gamma, y_raw, out if relu else None, mi)
        return out

    @staticmethod
    def backward(ctx, dout):
        return _ConvBN._backward_impl(ctx, dout)

    @staticmethod
    def _backward_impl(ctx, dout, want_dx=None, want_dw=None):
        """want_dx / want_dw: overrides of needs_input_grad[0] / [1] (_RootConvBN computes the input gradients itself and then
        gets (dx_raw, dw, dgamma, dbeta) back instead of the argument-ordered tuple)"""
        _p = _Args()
        x, weight, gamma, y_raw, out, mi = ctx.saved_tensors
        k, stride, pad, relu, training, has_res = ctx.cfg
        root = want_dx is not None
        need_dx = ctx.needs_input_grad[0] if want_dx is None else want_dx
        need_dw = ctx.needs_input_grad[1] if want_dw is None else want_dw
        if not training:
            raise _lib.CrError("backward through frozen BatchNorm is not implemented")
        Cout = weight.shape[0]
        dev = x.device
        dout = dout.to(x.dtype).contiguous()
        M = y_raw.numel() // Cout
        lib = _lib.load()
        sums = torch.empty((1025, 2, Cout), dtype=f32, device=dev)
        dx_raw = torch.empty_like(y_raw)
        dres = torch.empty_like(y_raw) if has_res else None
        gs, bs = grad_sink(gamma), grad_sink(ctx.beta_ref)
        if gs is not None and bs is not None:
            dgamma, dbeta, ret_g, ret_b = gs, bs, None, None
        else:
            dgamma = torch.zeros((Cout,), dtype=f32, device=dev)
            dbeta = torch.zeros((Cout,), dtype=f32, device=dev)
            ret_g, ret_b = dgamma, dbeta
        _chk(lib.cr_bn_bwd(_ctx(x), _p(dout), _p(out), _p(y_raw), _p(mi), _p(gamma.detach()), _p(sums), _p(dx_raw),
                           _p(dres), _p(dgamma), _p(dbeta), M, Cout, int(relu), _af(x)), "cr_bn_bwd")
        (xslot, xi), (rslot, ri) = ctx.slots
        if rslot is not None and dres is not None:          # the residual's other consumer (a convolution) adds this
            _slot_put(rslot, dres)
            dres = None
        dx = None
        if need_dx:
            _, wt = prepared_weights(weight, True, x.dtype)
            if xslot is not None and xi > 1:                # not the first consumer of x: leave the contribution in the slot
                _slot_put(xslot, conv_bwd_data_raw(dx_raw, wt, x.shape, k, stride, pad, accumulate=_slot_fold(xslot)))
            else:
                acc = _slot_take(xslot) if xslot is not None else None
                dx = conv_bwd_data_raw(dx_raw, wt, x.shape, k, stride, pad, accumulate=acc)
        elif xslot is not None:
            raise RuntimeError("gradient slot registered for an input that needs no gradient")
        dw = conv_bwd_weight_raw(dx_raw, x, k, stride, pad, grad_sink(weight)) if need_dw else None
        if root:
            return dx_raw, dw, ret_g, ret_b
        return dx, dw, ret_g, ret_b, dres, None, None, None, None, None, None, None, None, None


class _RootConvBN(torch.autograd.Function):
    """DLA `Root` (dla.py:156-174): 1x1 convolution + BatchNorm (+ ReLU) over the channel concatenation of its children.
    Forward: the concatenation is made inside (one cat kernel), then exactly _ConvBN.  Backward: the gradient of every child
    is its OWN backward-data GEMM on the matching row slice of the transposed weights (rows = input channels: a slice is
    contiguous) -- no gradient of the concatenation, no narrow / copy per child, and a child that another convolution
    consumed first gets its share through that convolution's gradient slot instead of an autograd add."""

    @staticmethod
    def forward(ctx, weight, gamma, beta, running_mean, running_var, relu, eps, momentum, slots, *children):
        with torch.no_grad():
            x = torch.cat(children, 3)
        out = _ConvBN.forward(ctx, x, weight, gamma, beta, None, running_mean, running_var, 1, 0, relu, eps, momentum, True)
        ctx.child_slots = slots
        ctx.child_ch = [c.shape[3] for c in children]
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight = ctx.saved_tensors[0], ctx.saved_tensors[1]
        nchild = len(ctx.child_ch)
        need = ctx.needs_input_grad
        # BatchNorm backward + weight gradient exactly as _ConvBN, without a gradient for the concatenated input
        ctx.slots = ((None, 0), (None, 0))
        res = _ConvBN._backward_impl(ctx, dout, want_dx=False, want_dw=need[0])
        dx_raw, dw, ret_g, ret_b = res
        _, wt = prepared_weights(weight, True, x.dtype)
        outs, c0 = [], 0
        N, H, W, _ = x.shape
        for i, ci in enumerate(ctx.child_ch):
            g = None
            if need[9 + i]:
                slot, idx = ctx.child_slots[i]
                g = conv_bwd_data_raw(dx_raw, wt[c0:c0 + ci], (N, H, W, ci), 1, 1, 0,
                                      accumulate=_slot_fold(slot) if slot is not None else None)
                if slot is not None:
                    _slot_put(slot, g)                       # the child's first consumer (a convolution / pooling) adds it
                    g = None
            elif ctx.child_slots[i][0] is not None:
                raise RuntimeError("gradient slot registered for an input that needs no gradient")
            outs.append(g)
            c0 += ci
        return (dw, ret_g, ret_b, None, None, None, None, None, None) + tuple(outs)


def root_conv_bn_act(children, weight, gamma, beta, running_mean, running_var, relu=True, eps=1e-5, momentum=0.1, training=True):
    """conv_bn_act(torch.cat(children, 3), 1x1 weight, ...) for DLA's Root nodes in training mode (see _RootConvBN)"""
    children = list(children)
    if not (training and torch.is_grad_enabled() and _ROOT_FUSED[0] and all(c.dim() == 4 for c in children)
            and all(c.shape[3] % 16 == 0 for c in children) and weight.shape[2] == 1):
        return conv_bn_act(torch.cat(children, 3), weight, gamma, beta, running_mean, running_var, 1, 0, relu, None, eps,
                           momentum, training)
    slots = tuple(_slot_register(c, False) for c in children)
    return _RootConvBN.apply(as_krsc(weight), gamma, beta, running_mean, running_var, relu, eps, momentum, slots, *children)


_ROOT_FUSED = [os.environ.get("CR_ROOT_FUSED", "1") == "1"]
_STATS_EPOCH = [0]          # bumped by every eager train-mode BatchNorm forward (running statistics change through raw pointers)


_FOLD_REG = [None]          # a list while graphed.GraphedDenseEval captures: the folds the graph reads (see refresh_folds)


def refresh_folds(reg):
    """re-fold (cr_fold_bn, in place) every registered BatchNorm whose inputs changed since its buffers were written; the
    check is host-side (version counters, weight / statistics epochs, storage addresses)"""
    lib = None
    for e in reg:
        w, g, b, m, v = e["t"]
        tag = (w._version, g._version, b._version, m._version, v._version, _WEIGHT_EPOCH[0], _STATS_EPOCH[0], w.data_ptr(), m.data_ptr())
        if tag == e["tag"]:
            continue
        lib = lib or _lib.load()
        _chk(lib.cr_fold_bn(_ctx(e["wf"]), _p(w.detach()), _p(g.detach()), _p(b.detach()), _p(m), _p(v), e["eps"], _p(e["wf"]),
                            _p(e["b"]), e["Cout"], e["K"], e["af"]), "cr_fold_bn")
        e["tag"] = tag


def conv_bn_folded(x, weight, gamma, beta, running_mean, running_var, stride=1, pad=0, relu=True, residual=None, eps=1e-5):
    """inference: frozen BatchNorm folded into the convolution's weights and bias (cr_fold_bn), residual and ReLU in the
    conv epilogue -- one kernel per layer instead of conv + scale/shift arithmetic + a second pass over the output.
    Outside graph capture the folded copies are cached on the weight tensor (keyed by the version counters of the five
    tensors, the weight epoch and the statistics epoch); inside a capture they are recomputed, so that a replay follows
    weights that changed since."""
    _p = _Args()
    _need_cuda(x, "conv input")
    weight = as_krsc(weight)
    Cout, _, k, _ = weight.shape
    K_ = weight.numel() // Cout
    capturing = torch.cuda.is_current_stream_capturing()
    attr = "_cr_fold" if x.dtype == bf16 else "_cr_fold32"
    warm = getattr(weight, attr, None)
    if capturing and _FOLD_REG[0] is not None and warm is not None:
        # eval-mode graph with externally refreshed folds: the graph only READS the folded copies; refresh_folds() rewrites
        # them (eagerly, before a replay) when one of the five tensors or the weight / statistics epoch has moved.  The
        # buffers are the ones the warm-up pass allocated OUTSIDE the capture: a tensor allocated during the capture may share
        # its block with an earlier intermediate of the graph, which every replay rewrites.
        _, wf, bias_f = warm
        _FOLD_REG[0].append({"t": (weight, gamma, beta, running_mean, running_var), "eps": float(eps), "wf": wf, "b": bias_f,
                             "tag": None, "af": _af(x), "K": K_, "Cout": Cout})
        try:
            delattr(weight, attr)           # the graph owns these buffers now: eager calls make their own
        except Exception:
            pass
        return conv_fwd_raw(x, wf, Cout, k, stride, pad, bias=bias_f, residual=residual, relu=relu)
    tag = None if capturing else (weight._version, gamma._version, beta._version, running_mean._version, running_var._version,
                                  _WEIGHT_EPOCH[0], _STATS_EPOCH[0], weight.data_ptr(), running_mean.data_ptr(), float(eps))
    ent = None if capturing else warm
    if ent is None or ent[0] != tag:
        wf = torch.empty((Cout, K_), dtype=x.dtype, device=x.device)
        bias_f = torch.empty((Cout,), dtype=f32, device=x.device)
        wd, gd, bd = weight.detach(), gamma.detach(), beta.detach()
        _chk(_lib.load().cr_fold_bn(_ctx(x), _p(wd), _p(gd), _p(bd), _p(running_mean),
                                    _p(running_var), float(eps), _p(wf), _p(bias_f), Cout, K_, _af(x)), "cr_fold_bn")
        ent = (tag, wf, bias_f)
        if not capturing:
            try:
                setattr(weight, attr, ent)
            except Exception:
                pass
    return conv_fwd_raw(x, ent[1], Cout, k, stride, pad, bias=ent[2], residual=residual, relu=relu)


def conv_bn_act(x, weight, gamma, beta, running_mean, running_var, stride=1, pad=0, relu=True, residual=None,
                eps=1e-5, momentum=0.1, training=True):
    if not training and not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or gamma.requires_grad
                                                          or (residual is not None and residual.requires_grad))):
        return conv_bn_folded(x, weight, gamma, beta, running_mean, running_var, stride, pad, relu, residual, eps)
    xs = _slot_register(x, True)
    rs = _slot_register(residual, False) if residual is not None else (None, 0)
    return _ConvBN.apply(x, as_krsc(weight), gamma, beta, residual, running_mean, running_var, stride, pad, relu, eps,
                         momentum, training, (xs, rs))


# --------------------------------------------------------------------------
# conv + bias (+ ReLU), bf16 or f32 output   (FPN laterals/outputs, RPN head)
# --------------------------------------------------------------------------
def relu_bwd(y, dy):
    """dy masked by the ReLU whose OUTPUT is y (one kernel); dy is cast to y's dtype first if it differs"""
    _p = _Args()
    dy = _act_cast(dy, y.dtype)
    g = torch.empty_like(dy)
    _chk(_lib.load().cr_relu_bwd(_ctx(y), _p(y.contiguous()), _p(dy), _p(g), y.numel(), _af(y)), "cr_relu_bwd")
    return g


class _ConvBias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, relu, out_f32, slot=(None, 0)):
        ctx.slot = slot
        Cout, Cin, k, _ = weight.shape
        wb, _ = prepared_weights(weight, False, x.dtype)
        y = conv_fwd_raw(x, wb, Cout, k, stride, pad, bias=None if bias is None else bias.detach(), relu=relu,
                         out_f32=out_f32)
        ctx.cfg = (k, stride, pad, relu, bias is not None)
        ctx.bias_ref = bias
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        _p = _Args()
        x, weight, y = ctx.saved_tensors
        k, stride, pad, relu, has_bias = ctx.cfg
        g = dy.contiguous()
        if relu:
            g = relu_bwd(y, g)
        db, acc = None, None
        want_db = has_bias and ctx.needs_input_grad[2]
        if want_db:
            C = g.shape[3]
            bsink = grad_sink(ctx.bias_ref)
            acc = bsink if bsink is not None else torch.zeros((C,), dtype=f32, device=g.device)
            db = None if bsink is not None else acc
            if not (ctx.needs_input_grad[1] and g.dtype == x.dtype):
                # no weight-gradient launch to ride on (or an f32 upstream gradient): separate column sum
                ws = torch.empty((1024, C), dtype=f32, device=g.device)
                lib = _lib.load()
                _chk(lib.cr_colsum_accum(_ctx(g), _p(g), int(g.dtype == f32), g.numel() // C, C, _p(ws), _p(acc)),
                     "cr_colsum_accum")
                acc = None
        g = g.to(x.dtype).contiguous()
        dx = None
        xslot, xi = ctx.slot
        if ctx.needs_input_grad[0]:
            _, wt = prepared_weights(weight, True, x.dtype)
            if xslot is not None and xi > 1:
                _slot_put(xslot, conv_bwd_data_raw(g, wt, x.shape, k, stride, pad, accumulate=_slot_fold(xslot)))
            else:
                dx = conv_bwd_data_raw(g, wt, x.shape, k, stride, pad, accumulate=_slot_take(xslot) if xslot is not None else None)
        elif xslot is not None:
            raise RuntimeError("gradient slot registered for an input that needs no gradient")
        # the bias gradient (column sums of dy) is accumulated inside the weight-gradient kernel
        dw = conv_bwd_weight_raw(g, x, k, stride, pad, grad_sink(weight), bias_acc=acc) if ctx.needs_input_grad[1] else None
        return dx, dw, db, None, None, None, None, None


class _PadInputChannels(torch.autograd.Function):
    """(Cout,Cin,k,k) weight -> zero-padded to Cin8 input channels (the RGB stem: 3 -> 8).  Backward adds the slice of
    the padded gradient straight into the parameter's gradient sink when there is one."""
    @staticmethod
    def forward(ctx, w, cin_to):
        ctx.w_ref = w
        out = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, cin_to - w.shape[1]))
        return out.contiguous(memory_format=torch.channels_last)

    @staticmethod
    def backward(ctx, g):
        w = ctx.w_ref
        gs = g[:, :w.shape[1]]
        sink = grad_sink(w)
        if sink is not None:
            sink.add_(gs)
            return None, None
        return gs, None


def pad_input_channels(w, cin_to=8):
    return _PadInputChannels.apply(w, cin_to)


class _CatRows(torch.autograd.Function):
    """concatenate parameters along dim 0 (+ zero rows up to `rows`); backward routes each slice into its sink."""
    @staticmethod
    def forward(ctx, rows, *params):
        ctx.refs = params
        n = sum(p.shape[0] for p in params)
        parts = list(params)
        if rows > n:
            parts.append(params[0].new_zeros((rows - n,) + tuple(params[0].shape[1:])))
        return torch.cat(parts, 0)

    @staticmethod
    def backward(ctx, g):
        outs, off = [], 0
        for p in ctx.refs:
            gs = g[off:off + p.shape[0]]
            off += p.shape[0]
            sink = grad_sink(p)
            if sink is not None:
                sink.add_(gs)
                outs.append(None)
            else:
                outs.append(gs)
        return (None,) + tuple(outs)


def cat_rows(params, rows):
    return _CatRows.apply(rows, *params)


def conv_bias_act(x, weight, bias, stride=1, pad=0, relu=False, out_f32=False):
    return _ConvBias.apply(x, as_krsc(weight), bias, stride, pad, relu, out_f32, _slot_register(x, True))


# --------------------------------------------------------------------------
# grouped convolutions: the five pyramid levels of an FPN output conv / the RPN head conv in ONE launch per direction
# --------------------------------------------------------------------------
_GROUP_ON = [os.environ.get("CR_CONV_GROUP", "1") == "1"]


def _ptr_table(tensors):
    return (ctypes.c_void_p * len(tensors))(*[0 if t is None else t.data_ptr() for t in tensors])


def _int_table(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def group_supported(xs, weights, stride=1):
    """shapes the grouped kernels take: stride 1, k in {1,3}, Cout % 128 == 0, Cin % 64 == 0, fp32 or bf16 (not the split mode)"""
    w0 = weights[0]
    Cout, Cin, k, _ = w0.shape
    return (_GROUP_ON[0] and stride == 1 and 2 <= len(xs) <= 8 and k in (1, 3) and Cout % 128 == 0 and Cin % 128 == 0
            and all(tuple(w.shape) == tuple(w0.shape) for w in weights) and all(x.is_cuda and x.dim() == 4 for x in xs)
            and not (xs[0].dtype == f32 and _ACT[0][1] == 2))


def conv_fwd_group_raw(xs, wbs, ys, Cin, Cout, k, pad, biases, relu):
    """cr_conv2d_fwd_group on prepared operands (module-level so that bench.py can bracket the launch with events)"""
    cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
    _chk(_lib.load().cr_conv2d_fwd_group(
        _ctx(xs[0]), len(xs), cast(_ptr_table(xs)), cast(_ptr_table(wbs)), cast(_ptr_table(ys)), cast(_int_table([x.shape[0] for x in xs])),
        cast(_int_table([x.shape[1] for x in xs])), cast(_int_table([x.shape[2] for x in xs])), Cin, Cout, k, pad,
        cast(_ptr_table(biases)), None, int(relu), _af(xs[0])), "cr_conv2d_fwd_group")


def conv_bwd_data_group_raw(gs, wts, outs, shapes, Cin, Cout, k, pad, accs):
    cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
    _chk(_lib.load().cr_conv2d_bwd_data_group(
        _ctx(gs[0]), len(gs), cast(_ptr_table(gs)), cast(_ptr_table(wts)), cast(_ptr_table(outs)),
        cast(_int_table([s_[0] for s_ in shapes])), cast(_int_table([s_[1] for s_ in shapes])), cast(_int_table([s_[2] for s_ in shapes])),
        Cin, Cout, k, pad, _af(gs[0]), cast(_ptr_table(accs))), "cr_conv2d_bwd_data_group")


def conv_bwd_weight_group_raw(gs, xs, dws, dbs, Cin, Cout, k, pad):
    cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
    _chk(_lib.load().cr_conv2d_bwd_weight_group(
        _ctx(xs[0]), len(xs), cast(_ptr_table(gs)), cast(_ptr_table(xs)), cast(_ptr_table(dws)), cast(_ptr_table(dbs)),
        cast(_int_table([x.shape[0] for x in xs])), cast(_int_table([x.shape[1] for x in xs])), cast(_int_table([x.shape[2] for x in xs])),
        Cin, Cout, k, pad, 1), "cr_conv2d_bwd_weight_group")


# --------------------------------------------------------------------------
# Winograd F(2x2, 3x3) route of the grouped 3x3 convolutions (float32, stride 1, pad 1, even maps, one shared weight):
# input transform -> 16 GEMMs (two grouped 1x1 launches of 8) -> output transform.  2.25x fewer multiplies; the V / M planes
# (16 x tiles x C floats each) are persistent per (tiles, channels) and must exist before a graph capture (first eager pass).
# --------------------------------------------------------------------------
_WINO_BUF = {}


def winograd_on():
    return os.environ.get("CR_WINOGRAD", "1") == "1"


def _wino_buffers(T, C, O, dev):
    key = (T, C, O, str(dev))
    ent = _WINO_BUF.get(key)
    if ent is None:
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise _lib.CrError("winograd: the V / M planes must be allocated before the graph capture (run one eager pass first)")
        # U (16, O, C) is the head of a flat buffer whose 16 * O tail holds the partial bias gradients of the weight-gradient
        # route (one zero-fill covers both)
        ent = _WINO_BUF[key] = (torch.empty((16, T, C), dtype=f32, device=dev), torch.empty((16, T, O), dtype=f32, device=dev),
                                torch.empty((16 * O * C + 16 * O,), dtype=f32, device=dev))
    return ent


def wino_supported(xs, weight, k, pad):
    return (winograd_on() and k == 3 and pad == 1 and xs[0].dtype == f32 and _ACT[0][1] == 1 and len(xs) <= 8
            and weight.shape[0] % 128 == 0 and weight.shape[1] % 128 == 0       # (backward-data swaps the two)
            and all(x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0 for x in xs))


WINO_MIN_TILES = int(os.environ.get("CR_WINO_MIN_TILES", "4096"))


def _wino_plan(xs, ws, bs, k, pad):
    """"shared": every map through ONE Winograd pipeline (one weight and bias object for all); else per map True / False"""
    n = len(xs)
    if not wino_supported(xs, ws[0], k, pad):
        return [False] * n
    if all(w is ws[0] for w in ws) and all(b is bs[0] for b in bs):
        return "shared"
    return [x.shape[0] * (x.shape[1] // 2) * (x.shape[2] // 2) >= WINO_MIN_TILES for x in xs]


def wino_gemm_raw(V, U, M, T, cin, cout):
    """M[k] (T,cout) = V[k] (T,cin) @ U[k] (cout,cin)^T for the 16 transformed positions: one launch (module-level so that
    bench.py can bracket it with events)"""
    _p = _Args()
    _chk(_lib.load().cr_gemm_batched_f32(_ctx(V), _p(V), _p(U), _p(M), T, cin, cout, 16, T * cin, cout * cin, T * cout),
         "cr_gemm_batched_f32")


def wino_conv3x3_group(srcs, w_krsc, dsts, bias, relu, accs, backward, keep_v=False):
    """dsts[i] = conv3x3(srcs[i], w) (+ bias, ReLU, + accs[i]) for n maps sharing ONE weight: forward (w (O,3,3,C) applied to
    (N,H,W,C) maps) or, backward = True, the backward-data of that convolution (srcs = dY with O channels, dsts = dX with C).
    keep_v: the transformed input goes to a tensor of its own, which is returned (the weight gradient multiplies the same
    planes: wino_wgrad_group(v_saved=...)), instead of the shared planes the next Winograd convolution overwrites."""
    cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
    lib = _lib.load()
    _p = _Args()
    assert w_krsc.dim() == 4 and w_krsc.shape[2:] == (3, 3) and w_krsc.is_contiguous(memory_format=torch.channels_last)
    O, C = w_krsc.shape[0], w_krsc.shape[1]              # logical (O, C, 3, 3) over physical [O][3][3][C]
    cin, cout = (O, C) if backward else (C, O)          # channels of the maps going in / coming out
    dev = srcs[0].device
    T = sum(x.shape[0] * (x.shape[1] // 2) * (x.shape[2] // 2) for x in srcs)
    V, M, U = _wino_buffers(T, cin, cout, dev)
    U = U[:16 * O * C]
    if keep_v:
        V = torch.empty((16, T, cin), dtype=f32, device=dev)
    ctx = _ctx(srcs[0])
    _chk(lib.cr_wino_filter(ctx, _p(w_krsc), _p(U), O, C, int(backward)), "cr_wino_filter")
    Ns, Hs, Ws = _int_table([x.shape[0] for x in srcs]), _int_table([x.shape[1] for x in srcs]), _int_table([x.shape[2] for x in srcs])
    _chk(lib.cr_wino_input(ctx, len(srcs), cast(_ptr_table(srcs)), cast(Ns), cast(Hs), cast(Ws), cin, _p(V), T), "cr_wino_input")
    wino_gemm_raw(V, U, M, T, cin, cout)
    _chk(lib.cr_wino_output(ctx, len(dsts), _p(M), cast(_ptr_table(dsts)), cast(Ns), cast(Hs), cast(Ws), cout, T, _p(bias), int(relu),
                            cast(_ptr_table(accs)) if accs is not None else None), "cr_wino_output")
    return V if keep_v else None


def wino_wgrad_on():
    """the Winograd weight gradient (CR_WINO_WGRAD, default on) accumulates over pixel splits with f32 atomics: the
    bit-reproducible mode (CR_DETERMINISTIC=1) keeps the direct kernels with their fixed-order reduce"""
    return os.environ.get("CR_WINO_WGRAD", "1") != "0" and os.environ.get("CR_DETERMINISTIC", "0") in ("", "0")


def wino_wgrad_group(gs, xs, w_sink, b_sink, v_saved=None):
    """weight (and bias) gradient of ONE 3x3 weight applied to n maps, the Winograd way: dM = A dY A^T and V = B^T x B as for
    the forward pass, dU[k] = dM[k]^T V[k] per transformed position (cr_linear_bwd_weight), dW += G^T dU G into the flat
    gradient; db += channel sums of dY.  45.8 instead of 103 GFLOP for the five pyramid levels of 4 images."""
    cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
    lib = _lib.load()
    _p = _Args()
    O, C = gs[0].shape[3], xs[0].shape[3]
    dev = xs[0].device
    T = sum(x.shape[0] * (x.shape[1] // 2) * (x.shape[2] // 2) for x in xs)
    V, M, U = _wino_buffers(T, C, O, dev)
    ctx = _ctx(xs[0])
    Ns, Hs, Ws = _int_table([x.shape[0] for x in xs]), _int_table([x.shape[1] for x in xs]), _int_table([x.shape[2] for x in xs])
    if v_saved is not None:                 # the forward pass kept B^T x B of exactly these maps
        assert v_saved.shape == (16, T, C)
        V = v_saved
    else:
        _chk(lib.cr_wino_input(ctx, len(xs), cast(_ptr_table(xs)), cast(Ns), cast(Hs), cast(Ws), C, _p(V), T), "cr_wino_input")
    U.zero_()
    part = U[16 * O * C:] if b_sink is not None else None
    _chk(lib.cr_wino_dy(ctx, len(gs), cast(_ptr_table(gs)), cast(Ns), cast(Hs), cast(Ws), O, _p(M), T, _p(part)), "cr_wino_dy")
    _chk(lib.cr_wgrad_batched_f32(ctx, _p(M), _p(V), _p(U), T, C, O, 16, T * O, T * C, O * C), "cr_wgrad_batched_f32")
    _chk(lib.cr_wino_filter_grad(ctx, _p(U), _p(w_sink), O, C, _p(part), _p(b_sink)), "cr_wino_filter_grad")


class _ConvBiasGroup(torch.autograd.Function):
    """y_i = act(conv(x_i, W_i) + b_i) for n same-geometry problems: cr_conv2d_fwd_group forward, cr_conv2d_bwd_data_group and
    (fp32) cr_conv2d_bwd_weight_group backward.  The W_i / b_i may be one parameter repeated (the RPN head)."""

    @staticmethod
    def forward(ctx, n, pad, relu, slots, *args):
        """slots[-1] == "stacked" (conv_bias_act_group(stacked=True)): the n outputs are the consecutive row blocks of ONE
        (1, 1, sum of pixels, Cout) tensor -- what a following 1x1 convolution treats as a single map (the RPN predictors
        over all pyramid levels at once); one gradient comes back and one ReLU backward covers all levels."""
        stacked = len(slots) == n + 1
        xs, ws, bs = args[:n], args[n:2 * n], args[2 * n:3 * n]
        _p = _Args()
        Cout, Cin, k, _ = ws[0].shape
        dt = xs[0].dtype
        xs = [x.contiguous() for x in xs]
        wbs = [prepared_weights(w, False, dt)[0] for w in ws]
        shapes = [(x.shape[0], x.shape[1] + 2 * pad - k + 1, x.shape[2] + 2 * pad - k + 1, Cout) for x in xs]
        if stacked:
            rows = [s[0] * s[1] * s[2] for s in shapes]
            Y = torch.empty((1, 1, sum(rows), Cout), dtype=dt, device=xs[0].device)
            flat, off, ys = Y.view(-1, Cout), 0, []
            for s, m in zip(shapes, rows):
                ys.append(flat[off:off + m].view(s))
                off += m
        else:
            ys = [torch.empty(s, dtype=dt, device=x.device) for s, x in zip(shapes, xs)]
        bd = [None if b is None else b.detach() for b in bs]
        # float32 3x3: the Winograd route -- all maps in one pipeline when they share the weight (the RPN head), the big maps
        # one by one otherwise (the FPN output convolutions; the small levels stay one direct grouped launch)
        ctx.wino = plan = _wino_plan(xs, ws, bs, k, pad)
        # the transformed input is kept for the weight gradient when that will take the Winograd route too
        keep = lambda i: (ctx.needs_input_grad[4 + n + i] and grad_sink(ws[i]) is not None and wino_wgrad_on()
                          and os.environ.get("CR_WINO_KEEP_V", "1") != "0")
        ctx.wino_v = {}
        if plan == "shared":
            ctx.wino_v["shared"] = wino_conv3x3_group(xs, wbs[0], ys, bd[0], relu, None, False, keep(0))
        else:
            for i in range(n):
                if plan[i]:
                    ctx.wino_v[i] = wino_conv3x3_group([xs[i]], wbs[i], [ys[i]], bd[i], relu, None, False, keep(i))
            rest = [i for i in range(n) if not plan[i]]
            if rest:
                conv_fwd_group_raw([xs[i] for i in rest], [wbs[i] for i in rest], [ys[i] for i in rest], Cin, Cout, k, pad,
                                   [bd[i] for i in rest], relu)
        ctx.cfg = (n, k, pad, relu)
        ctx.slots = slots
        ctx.refs = (ws, bs)
        ctx.stacked = shapes if stacked else None
        ctx.save_for_backward(*xs, *((Y,) if stacked and relu else (ys if relu else ())))
        ctx.set_materialize_grads(False)
        return Y if stacked else tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        n, k, pad, relu = ctx.cfg
        ws, bs = ctx.refs
        saved = ctx.saved_tensors
        xs = saved[:n]
        _p = _Args()
        lib = _lib.load()
        cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
        gs = {}
        if ctx.stacked is not None:
            live = list(range(n)) if dys[0] is not None else []
            if live:
                g = dys[0].contiguous()
                if relu:
                    g = relu_bwd(saved[n], g)
                g = g.to(xs[0].dtype).contiguous().view(-1, g.shape[-1])
                off = 0
                for i, s in enumerate(ctx.stacked):
                    m = s[0] * s[1] * s[2]
                    gs[i] = g[off:off + m].view(s)
                    off += m
        else:
            ys = saved[n:] if relu else (None,) * n
            live = [i for i in range(n) if dys[i] is not None]            # an output nobody differentiated takes no part
            for i in live:
                g = dys[i].contiguous()
                if relu:
                    g = relu_bwd(ys[i], g)
                gs[i] = g.to(xs[i].dtype).contiguous()
        dt = xs[0].dtype
        Cout, Cin = ws[0].shape[0], ws[0].shape[1]
        dxs = [None] * n
        need_dx = [i for i in live if ctx.needs_input_grad[4 + i]]
        for i in range(n):
            if i not in need_dx and ctx.slots[i][0] is not None and i in live:
                raise RuntimeError("gradient slot registered for an input that needs no gradient")
        if need_dx:
            plan = ctx.wino
            outs = [torch.empty_like(xs[i]) for i in need_dx]
            accs = []
            for i in need_dx:
                slot, xi = ctx.slots[i]
                a = None if slot is None else (_slot_take(slot) if xi <= 1 else _slot_fold(slot))
                accs.append(None if a is None else a.to(dt).contiguous())
            if plan == "shared":
                wino_conv3x3_group([gs[i] for i in need_dx], prepared_weights(ws[0], False, dt)[0], outs, None, False, accs, True)
            else:
                for j, i in enumerate(need_dx):
                    if plan[i]:
                        wino_conv3x3_group([gs[i]], prepared_weights(ws[i], False, dt)[0], [outs[j]], None, False, [accs[j]], True)
                rest = [j for j, i in enumerate(need_dx) if not plan[i]]
                if rest:
                    conv_bwd_data_group_raw([gs[need_dx[j]] for j in rest], [prepared_weights(ws[need_dx[j]], True, dt)[1] for j in rest],
                                            [outs[j] for j in rest], [xs[need_dx[j]].shape for j in rest], Cin, Cout, k, pad,
                                            [accs[j] for j in rest])
            for i, o in zip(need_dx, outs):
                slot, xi = ctx.slots[i]
                if slot is not None and xi > 1:
                    _slot_put(slot, o)
                else:
                    dxs[i] = o
        # weight / bias gradients: one grouped launch in fp32 when every parameter has a sink to accumulate into; otherwise the
        # per-problem kernels (bf16 mode, parameters without sinks)
        dws, dbs = [None] * n, [None] * n
        wl = [i for i in live if ctx.needs_input_grad[4 + n + i]]
        sinks_ok = dt == f32 and all(grad_sink(ws[i]) is not None and (bs[i] is None or grad_sink(bs[i]) is not None) for i in wl)
        wplan = ctx.wino if sinks_ok and wino_wgrad_on() else None
        if wl and wplan == "shared":
            # one weight over all maps: dU[k] = dM[k]^T V[k] for the 16 transformed positions in one batched launch
            want_db = bs[0] is not None and ctx.needs_input_grad[4 + 2 * n + wl[0]]
            wino_wgrad_group([gs[i] for i in wl], [xs[i] for i in wl], grad_sink(ws[0]), grad_sink(bs[0]) if want_db else None,
                             ctx.wino_v.get("shared") if len(wl) == n else None)
            wl = []
        elif wl and wplan is not None and any(wplan[i] for i in wl):
            for i in [i for i in wl if wplan[i]]:
                want_db = bs[i] is not None and ctx.needs_input_grad[4 + 2 * n + i]
                wino_wgrad_group([gs[i]], [xs[i]], grad_sink(ws[i]), grad_sink(bs[i]) if want_db else None, ctx.wino_v.get(i))
            wl = [i for i in wl if not wplan[i]]
        if wl and sinks_ok and len(wl) > 1:
            dwt = [grad_sink(ws[i]) for i in wl]
            dbt = [None if bs[i] is None or not ctx.needs_input_grad[4 + 2 * n + i] else grad_sink(bs[i]) for i in wl]
            conv_bwd_weight_group_raw([gs[i] for i in wl], [xs[i] for i in wl], dwt, dbt, Cin, Cout, k, pad)
        else:
            for i in wl:
                bsink = None if bs[i] is None else grad_sink(bs[i])
                want_db = bs[i] is not None and ctx.needs_input_grad[4 + 2 * n + i]
                acc = None
                if want_db:
                    acc = bsink if bsink is not None else torch.zeros((Cout,), dtype=f32, device=xs[i].device)
                    if bsink is None:
                        dbs[i] = acc if dbs[i] is None else dbs[i] + acc
                dws[i] = conv_bwd_weight_raw(gs[i], xs[i], k, 1, pad, grad_sink(ws[i]), bias_acc=acc)
        return (None, None, None, None) + tuple(dxs) + tuple(dws) + tuple(dbs)


def conv_bias_act_group(xs, weights, biases, pad=0, relu=False, stacked=False):
    """[act(conv(x_i, W_i) + b_i)] -- one launch per direction when the shapes allow (group_supported), else one conv each.
    stacked=True (only when group_supported): ONE tensor (1, 1, sum_i N_i H_i W_i, Cout) whose row blocks are the outputs."""
    xs, weights, biases = list(xs), [as_krsc(w) for w in weights], list(biases)
    if not group_supported(xs, weights):
        if stacked:
            raise _lib.CrError("conv_bias_act_group(stacked=True): the problems do not form a group (see group_supported)")
        return [conv_bias_act(x, w, b, 1, pad, relu=relu) for x, w, b in zip(xs, weights, biases)]
    slots = tuple(_slot_register(x, True) for x in xs)
    if stacked:
        return _ConvBiasGroup.apply(len(xs), pad, relu, slots + ("stacked",), *xs, *weights, *biases)
    return list(_ConvBiasGroup.apply(len(xs), pad, relu, slots, *xs, *weights, *biases))


# --------------------------------------------------------------------------
# pooling / FPN top-down
# --------------------------------------------------------------------------
class _Pool2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, window, slot=(None, 0)):
        ctx.slot = slot
        _p = _Args()
        _need_cuda(x, "pool input")
        N, H, W, C = x.shape
        y = torch.empty((N, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
        lib = _lib.load()
        _chk(lib.cr_pool2x_fwd(_ctx(x), _p(x), _p(y), N, H, W, C, window, _af(x)), "cr_pool2x_fwd")
        ctx.window = window
        ctx.save_for_backward(x)
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    def backward(ctx, dy):
        _p = _Args()
        (x,) = ctx.saved_tensors
        N, H, W, C = x.shape
        xslot, xi = ctx.slot
        if dy is None:
            # every consumer of the pooled map handed its gradient on through gradient slots: nothing arrives here
            if xslot is None:
                return None, None, None
            if xi > 1:
                return None, None, None                      # what the slot holds stays for the accumulating consumer
            acc = _slot_take(xslot)
            return (None if acc is None else acc.to(x.dtype)), None, None
        dx = torch.empty_like(x)
        lib = _lib.load()
        dy = dy.to(x.dtype).contiguous()
        # the FIRST registered consumer of x (DLA: a level's input is pooled before its first convolution sees it) collects
        # what the others left in the slot, a later one folds the slot's content into its own result: added in the kernel
        acc = None if xslot is None else (_slot_take(xslot) if xi <= 1 else _slot_fold(xslot))
        acc = None if acc is None else acc.to(x.dtype).contiguous()
        _chk(lib.cr_pool2x_bwd(_ctx(x), _p(x), _p(dy), _p(dx), N, H, W, C, ctx.window, _af(x), _p(acc)), "cr_pool2x_bwd")
        if xslot is not None and xi > 1:
            _slot_put(xslot, dx)
            dx = None
        return dx, None, None


class _Pool3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _p = _Args()
        _need_cuda(x, "pool input")
        N, H, W, C = x.shape
        y = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=x.dtype, device=x.device)
        lib = _lib.load()
        _chk(lib.cr_maxpool3x3s2_fwd(_ctx(x), _p(x), _p(y), N, H, W, C, _af(x)), "cr_maxpool3x3s2_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        _p = _Args()
        (x,) = ctx.saved_tensors
        N, H, W, C = x.shape
        dx = torch.empty_like(x)
        lib = _lib.load()
        dy = dy.to(x.dtype).contiguous()
        _chk(lib.cr_maxpool3x3s2_bwd(_ctx(x), _p(x), _p(dy), _p(dx), N, H, W, C, _af(x)), "cr_maxpool3x3s2_bwd")
        return dx


def maxpool3x3s2(x):
    """nn.MaxPool2d(3, stride=2, padding=1) -- the torchvision ResNet stem (resnet.py:33,49)."""
    return _Pool3s2.apply(x)


def maxpool2x2(x):
    """nn.MaxPool2d(2, stride=2) -- dla.py:208."""
    return _Pool2x.apply(x, 2, _slot_register(x, True))


def subsample2x(x):
    """F.max_pool2d(kernel_size=1, stride=2) -- dla.py:474."""
    return _Pool2x.apply(x, 1, _slot_register(x, True))


class _UpsampleAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lat, top, slot=(None, 0)):
        ctx.slot = slot
        _p = _Args()
        _need_cuda(lat, "upsample_add input")
        N, H, W, C = lat.shape
        assert tuple(top.shape) == (N, H // 2, W // 2, C)
        y = torch.empty_like(lat)
        lib = _lib.load()
        assert lat.dtype == top.dtype
        _chk(lib.cr_upsample2x_add(_ctx(lat), _p(lat), _p(top), _p(y), N, H, W, C, _af(lat)), "cr_upsample2x_add")
        ctx.shape = (N, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        _p = _Args()
        N, H, W, C = ctx.shape
        dy = dy.contiguous()
        dtop = torch.empty((N, H // 2, W // 2, C), dtype=dy.dtype, device=dy.device)
        lib = _lib.load()
        _chk(lib.cr_sum2x2(_ctx(dy), _p(dy), _p(dtop), N, H, W, C, _af(dy)), "cr_sum2x2")
        tslot, ti = ctx.slot
        if tslot is not None:                                # the coarser level's output convolution adds this (see _GradSlot)
            _slot_put(tslot, dtop)
            dtop = None
        return dy, dtop, None


def upsample2x_add(lat, top):
    """detectron2 FPN top-down step: lateral + F.interpolate(top, scale_factor=2, mode='nearest')."""
    return _UpsampleAdd.apply(lat, top, _slot_register(top, False))


def preprocess(images_u8, mean, std, dtype=None):
    """(N,3,H,W) uint8 -> normalised NHWC with 8 (bf16) or 4 (f32) channels (3 real + zeros) in the activation dtype of the process
    (set_precision) unless `dtype` says otherwise: this call decides the precision of everything downstream."""
    _p = _Args()
    _need_cuda(images_u8, "images")
    assert images_u8.dtype == torch.uint8 and images_u8.is_contiguous()
    N, _, H, W = images_u8.shape
    dt = dtype if dtype is not None else act_dtype()
    y = torch.empty((N, H, W, 8 if dt == bf16 else 4), dtype=dt, device=images_u8.device)     # one 16-B chunk per pixel
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    lib = _lib.load()
    _chk(lib.cr_preprocess(_ctx(images_u8), _p(images_u8), _p(y), N, H, W, ctypes.cast(m, ctypes.c_void_p),
                           ctypes.cast(s, ctypes.c_void_p), _af(y)), "cr_preprocess")
    return y


# --------------------------------------------------------------------------
# ROIAlign over the FPN pyramid
# --------------------------------------------------------------------------
def _pyr_args(feats, scales):
    n = len(feats)
    ptrs = (ctypes.c_void_p * n)(*[f.data_ptr() for f in feats])
    Hs = (ctypes.c_int * n)(*[f.shape[1] for f in feats])
    Ws = (ctypes.c_int * n)(*[f.shape[2] for f in feats])
    sc = (ctypes.c_float * n)(*[float(s) for s in scales])
    cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
    return n, ptrs, Hs, Ws, sc, cast


_ROI_BWD_TILES = [os.environ.get("CR_ROI_BWD_TILES", "1") == "1"]


class _ROIAlign(torch.autograd.Function):
    """Where the pyramid gradient goes (backward), per level, in this order of preference:
      * a gradient slot of the level's map (the RPN head's convolution consumed it first, in the same autograd graph: its
        backward-data adds this contribution in its epilogue -- no fan-in add kernel);
      * the static gradient buffer of a captured dense region (`_cr_grad_dst` on the region's output, GraphedDense): the
        scatter lands where the backward graph reads it -- no copy;
      * a fresh zero-filled buffer returned to autograd."""

    @staticmethod
    def forward(ctx, rois, scales, out_size, slots, *feats):
        _p = _Args()
        _need_cuda(rois, "rois")
        C = feats[0].shape[3]
        R = rois.shape[0]
        dt = feats[0].dtype
        assert all(f.dtype == dt and f.is_contiguous() for f in feats)
        out = torch.empty((R, out_size, out_size, C), dtype=dt, device=rois.device)
        n, ptrs, Hs, Ws, sc, cast = _pyr_args(feats, scales)
        lib = _lib.load()
        rois = rois.contiguous()
        _chk(lib.cr_roi_align_fwd(_ctx(rois), cast(ptrs), cast(Hs), cast(Ws), cast(sc), n, C, _p(rois), R, out_size,
                                  out_size, _p(out), _af(out)), "cr_roi_align_fwd")
        ctx.cfg = (scales, out_size, [tuple(f.shape) for f in feats], dt)
        ctx.slots = slots
        # claimed (popped): a second RoIAlign over the same maps takes the ordinary path and autograd adds the two
        ctx.dsts = [f.__dict__.pop("_cr_grad_dst", None) if f.requires_grad else None for f in feats]
        ctx.save_for_backward(rois)
        return out

    @staticmethod
    def backward(ctx, dout):
        _p = _Args()
        (rois,) = ctx.saved_tensors
        scales, out_size, shapes, dt = ctx.cfg
        C = shapes[0][3]
        dsts = ctx.dsts
        # tile-owner kernel (cr_roi_align_bwd_set): writes every pixel of every level once, no atomics, bit-reproducible,
        # and the maps need no zero fill.  CR_ROI_BWD_TILES=0: the separable atomic kernel (A/B).
        tiles = _ROI_BWD_TILES[0] and out_size == 7 and C % 64 == 0 and len({s[0] for s in shapes}) == 1
        if all(d is not None and d.dtype == f32 and tuple(d.shape) == tuple(s) for d, s in zip(dsts, shapes)):
            grads = dsts
            if not tiles:
                torch._foreach_zero_(grads)
        else:
            # one buffer for the whole pyramid (the levels are views of it, each 16-B aligned); zero-filled for the atomics
            sizes = [(int(s[0] * s[1] * s[2] * s[3]) + 3) // 4 * 4 for s in shapes]
            flat = (torch.empty if tiles else torch.zeros)((sum(sizes),), dtype=f32, device=rois.device)
            grads, off = [], 0
            for s, n_ in zip(shapes, sizes):
                grads.append(flat[off:off + int(s[0] * s[1] * s[2] * s[3])].view(s))
                off += n_
        n, ptrs, Hs, Ws, sc, cast = _pyr_args(grads, scales)
        lib = _lib.load()
        dout = dout.to(dt).contiguous()
        if tiles:
            _chk(lib.cr_roi_align_bwd_set(_ctx(rois), cast(ptrs), cast(Hs), cast(Ws), cast(sc), n, C, int(shapes[0][0]), _p(rois),
                                          rois.shape[0], out_size, out_size, _p(dout), _af(dout)), "cr_roi_align_bwd_set")
        else:
            _chk(lib.cr_roi_align_bwd(_ctx(rois), cast(ptrs), cast(Hs), cast(Ws), cast(sc), n, C, _p(rois), rois.shape[0],
                                      out_size, out_size, _p(dout), _af(dout)), "cr_roi_align_bwd")
        res = []
        for g, (slot, _i) in zip(grads, ctx.slots):
            g = g if dt == f32 else g.to(dt)
            if slot is not None:
                _slot_put(slot, g)
                g = None
            res.append(g)
        return (None, None, None, None) + tuple(res)


class _SharedPrefix(torch.autograd.Function):
    """(B*S, ...) pooled RoI features -> (the same tensor, a copy of the first kf RoIs of every image).  The 3D head pools
    exactly the foreground slots the box head already pooled, so they are pooled ONCE; backward adds the 3D head's gradient
    into the box head's gradient tile in place (one small kernel) instead of scattering 20 % more RoIs with atomics."""

    @staticmethod
    def forward(ctx, pooled, B, S, kf):
        ctx.dims = (B, S, kf)
        v = pooled.view(B, S, -1)
        return pooled.view_as(pooled), v[:, :kf].reshape((B * kf,) + tuple(pooled.shape[1:]))

    @staticmethod
    def backward(ctx, g_all, g_head):
        B, S, kf = ctx.dims
        if g_all is None:
            g_all = torch.zeros((B * S,) + tuple(g_head.shape[1:]), dtype=g_head.dtype, device=g_head.device)
        g_all = g_all.contiguous()
        if g_head is not None:
            g_all.view(B, S, -1)[:, :kf] += g_head.reshape(B, kf, -1).to(g_all.dtype)
        return g_all, None, None, None


def shared_prefix(pooled, B, S, kf):
    return _SharedPrefix.apply(pooled, B, S, kf)


def roi_align_pyramid(feats, rois, scales, out_size):
    """feats: list of NHWC maps (fine -> coarse) in the activation dtype; rois (R,5) f32 [batch,x1,y1,x2,y2]."""
    slots = tuple(_slot_register(f, False) for f in feats)
    return _ROIAlign.apply(rois.to(f32), tuple(scales), out_size, slots, *feats)


# --------------------------------------------------------------------------
# fused Cube R-CNN decode + corner losses (K15/K16)
# --------------------------------------------------------------------------
class _CubeLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dxy, zr, dr, Ra, u, consts, flags):
        _p = _Args()
        _need_cuda(dxy, "cube head outputs")
        n = dxy.shape[0]
        ins = [t.contiguous().to(f32) for t in (dxy, zr, dr, Ra.reshape(n, 9), u)] + \
              [t.contiguous().to(f32) for t in consts]
        arr = (ctypes.c_void_p * 13)(*[t.data_ptr() for t in ins])
        losses = torch.empty((n, 5), dtype=f32, device=dxy.device)
        dec = torch.empty((n, 17), dtype=f32, device=dxy.device)
        lib = _lib.load()
        _chk(lib.cr_cube_loss_fwd(_ctx(dxy), ctypes.cast(arr, ctypes.c_void_p), n, *flags, _p(losses), _p(dec)),
             "cr_cube_loss_fwd")
        ctx.ins, ctx.flags = ins, flags
        ctx.set_materialize_grads(False)       # outputs nobody differentiates arrive as None, not as zero fills
        ctx.mark_non_differentiable(dec)
        return losses, dec

    @staticmethod
    def backward(ctx, gl, _gdec):
        _p = _Args()
        ins, flags = ctx.ins, ctx.flags
        n = ins[0].shape[0]
        dev = ins[0].device
        arr = (ctypes.c_void_p * 13)(*[t.data_ptr() for t in ins])
        g_dxy = torch.empty((n, 2), dtype=f32, device=dev)
        g_zr = torch.empty((n,), dtype=f32, device=dev)
        g_dr = torch.empty((n, 3), dtype=f32, device=dev)
        g_Ra = torch.empty((n, 3, 3), dtype=f32, device=dev)
        g_u = torch.empty((n,), dtype=f32, device=dev)
        lib = _lib.load()
        _chk(lib.cr_cube_loss_bwd(_ctx(ins[0]), ctypes.cast(arr, ctypes.c_void_p), n, *flags, _p(gl.contiguous()),
                                  _p(g_dxy), _p(g_zr), _p(g_dr), _p(g_Ra), _p(g_u)), "cr_cube_loss_bwd")
        return g_dxy, g_zr, g_dr, g_Ra, g_u, None, None


def cube_decode_loss(dxy, zr, dr, Ra, u, src_boxes, K4, v2r, prior_mean, gt2d, gtz, gtdims, gtR, allocentric=True,
                     chamfer_pose=True, use_conf=True, joint=True):
    """fused K15/K16 (see cr_cube_loss_fwd).  Returns (losses (n,5) [dims,xy,z,pose,joint], dec (n,17))."""
    consts = (src_boxes, K4, v2r, prior_mean, gt2d, gtz, gtdims, gtR.reshape(-1, 9))
    flags = (int(bool(allocentric)), int(bool(chamfer_pose)), int(bool(use_conf)), int(bool(joint)))
    return _CubeLoss.apply(dxy, zr, dr, Ra, u, consts, flags)


# --------------------------------------------------------------------------
# FC layers of the RoI heads on the hand-written implicit-GEMM kernels (cr_linear_*: a linear layer over R rows is a 1x1
# convolution over a (1,1,R,K) map), in both precision modes -- no library GEMM on the path.
# --------------------------------------------------------------------------
def prepared_fc_weight(weight, chw=None, dtype=bf16, need_transposed=False):
    """compute copies of an nn.Linear weight (O, K) f32 in `dtype`, cached per weight epoch on the tensor:
    wp (O,K) and, when asked, wt (K,O) for the backward-data GEMM.  chw = (C,H,W): the columns are re-ordered from the
    checkpoint's (c,h,w) flattening to (h,w,c) for NHWC-flattened RoI features.  f32 without a permutation: wp is the
    weight itself."""
    _p = _Args()
    attr = "_cr_fccache" if dtype == bf16 else "_cr_fccache32"
    ent = getattr(weight, attr, None)
    tag = (weight._version, _WEIGHT_EPOCH[0], weight.data_ptr(), chw)
    lib = _lib.load()
    O, K = weight.shape
    if ent is None or ent[0] != tag:
        if dtype == f32 and chw is None:
            wp = weight.detach()
        else:
            C, HW = (chw[0], chw[1] * chw[2]) if chw is not None else (K, 1)
            assert C * HW == K
            wp = torch.empty((O, K), dtype=dtype, device=weight.device)
            _chk(lib.cr_fc_weight_prepare(_ctx(weight), _p(weight.detach().contiguous()), _p(wp), O, C, HW, int(dtype == f32)),
                 "cr_fc_weight_prepare")
        ent = [tag, wp, None]
        try:
            setattr(weight, attr, ent)
        except Exception:
            pass
    if need_transposed and ent[2] is None:
        wt = torch.empty((K, O), dtype=dtype, device=weight.device)
        _chk(lib.cr_transpose2d(_ctx(weight), _p(ent[1].contiguous()), _p(wt), O, K, int(dtype == f32)), "cr_transpose2d")
        ent[2] = wt
    return (ent[1], ent[2]) if need_transposed else ent[1]


def _act_cast(t, dtype):
    """f32 tensor -> the activation dtype (own cast kernel for bf16; no copy for f32)"""
    if t.dtype == dtype:
        return t.contiguous()
    if dtype == bf16 and t.dtype == f32:
        _p = _Args()
        tc = t.contiguous()
        out = torch.empty(tc.shape, dtype=bf16, device=t.device)
        _chk(_lib.load().cr_cast_f32_to_bf16(_ctx(t), _p(tc), _p(out), tc.numel()), "cr_cast_f32_to_bf16")
        return out
    return t.to(dtype).contiguous()


def linear_fwd_raw(x, w, bias, out_f32=False, relu=False):
    """y (R,O) = x (R,K) @ w (O,K)^T + bias on the MFMA; x / w in the same activation dtype, bias f32 or None"""
    _p = _Args()
    af = _af(x)
    R, K = x.shape
    O = w.shape[0]
    if O % 16 or K % 16:
        raise _lib.CrError(f"linear: the GEMM kernels need O and K multiples of 16 (got {O}x{K}); use linear_cat for predictors")
    assert w.dtype == x.dtype and x.is_contiguous() and w.is_contiguous()
    y = torch.empty((R, O), dtype=f32 if (out_f32 or af) else bf16, device=x.device)
    _chk(_lib.load().cr_linear_fwd(_ctx(x), _p(x), _p(w), _p(bias), _p(y), R, K, O, int(relu), int(out_f32), af,
                                   _p(_w3(w, af))), "cr_linear_fwd")
    return y


def linear_bwd_data_raw(dy, wt):
    _p = _Args()
    R, O = dy.shape
    K = wt.shape[0]
    assert wt.dtype == dy.dtype and dy.is_contiguous() and wt.is_contiguous()
    dx = torch.empty((R, K), dtype=dy.dtype, device=dy.device)
    af = _af(dy)
    _chk(_lib.load().cr_linear_bwd_data(_ctx(dy), _p(dy), _p(wt), _p(dx), R, K, O, af, _p(_w3(wt, af))), "cr_linear_bwd_data")
    return dx


def linear_bwd_weight_raw(dy, x, dw, dbias, accumulate):
    """dw (O,K) f32 (+)= dy^T x; dbias (O) f32 += column sums of dy (None: not wanted)"""
    _p = _Args()
    R, O = dy.shape
    K = x.shape[1]
    assert dy.dtype == x.dtype and dy.is_contiguous() and x.is_contiguous() and dw.dtype == f32
    _chk(_lib.load().cr_linear_bwd_weight(_ctx(dy), _p(dy), _p(x), _p(dw), _p(dbias), R, K, O, int(accumulate), _af(x)),
         "cr_linear_bwd_weight")


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, chw, relu=False):
        _need_cuda(x, "linear input")
        dt = x.dtype
        wp = prepared_fc_weight(weight, chw, dt)
        xc = x.contiguous()
        y = linear_fwd_raw(xc, wp, None if bias is None else bias.detach(), relu=relu)
        ctx.save_for_backward(xc, y if relu else None)
        ctx.refs = (weight, bias, chw)
        return y

    @staticmethod
    def backward(ctx, dy):
        _p = _Args()
        xc, yr = ctx.saved_tensors
        weight, bias, chw = ctx.refs
        dt = xc.dtype
        lib = _lib.load()
        O, K = weight.shape
        C, HW = (chw[0], chw[1] * chw[2]) if chw is not None else (K, 1)
        dy = _act_cast(dy, dt)
        if yr is not None:
            dy = relu_bwd(yr, dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            _, wt = prepared_fc_weight(weight, chw, dt, need_transposed=True)
            dx = linear_bwd_data_raw(dy, wt)
        want_db = bias is not None and ctx.needs_input_grad[2]
        bacc = None
        if want_db:
            bacc = grad_sink(bias)
            if bacc is None:
                bacc = torch.zeros((O,), dtype=f32, device=dy.device)
                db = bacc
        if ctx.needs_input_grad[1]:
            acc = grad_sink(weight)
            if HW == 1 and acc is not None:
                # weight AND bias gradients land in the flat gradient straight from the GEMM
                linear_bwd_weight_raw(dy, xc, acc, bacc, accumulate=True)
            else:
                g = torch.empty((O, K), dtype=f32, device=dy.device)     # in the compute (h,w,c) column order
                linear_bwd_weight_raw(dy, xc, g, bacc, accumulate=False)
                if acc is None:
                    acc = torch.zeros((O, K), dtype=f32, device=dy.device)
                    dw = acc
                _chk(lib.cr_fc_grad_accum(_ctx(dy), _p(g), _p(acc), O, C, HW, 1), "cr_fc_grad_accum")
        elif want_db:
            ws = torch.empty((1024, O), dtype=f32, device=dy.device)
            _chk(lib.cr_colsum_accum(_ctx(dy), _p(dy), int(dt == f32), dy.shape[0], O, _p(ws), _p(bacc)), "cr_colsum_accum")
        return dx, dw, db, None, None


def linear(x, weight, bias=None, chw=None, relu=False):
    """F.linear on the MFMA in x's precision (own implicit-GEMM kernels) with the weight copies cached per optimizer step
    and the gradients accumulated straight into the optimizer's flat gradient (when the parameters carry sinks).
    chw: see prepared_fc_weight.  Needs O % 16 == 0 (see linear_cat for predictors with odd widths)."""
    return _Linear.apply(x, weight, bias, chw, relu)


class _CatPlan:
    """persistent buffers of one head's stacked predictors: W (Op,K) / b (Op) f32, refreshed from the parameters by ONE
    launch per weight epoch; the transposed / split copies for backward-data; dW / db that receive the stacked gradient;
    the segment tables that move rows between the parameters (or their gradient sinks) and the stacks."""

    def __init__(self, weights, biases):
        import numpy as np
        dev = weights[0].device
        self.sizes = [int(w.shape[0]) for w in weights]
        self.K = K = int(weights[0].shape[1])
        O = sum(self.sizes)
        self.Op = Op = (O + 15) // 16 * 16
        self.W = torch.zeros((Op, K), dtype=f32, device=dev)
        self.b = torch.zeros((Op,), dtype=f32, device=dev)
        self.dW = torch.empty((Op, K), dtype=f32, device=dev)
        self.db = torch.empty((Op,), dtype=f32, device=dev)
        self.sinks = all(grad_sink(t) is not None for t in list(weights) + list(biases))
        dt = np.dtype([("src", "<u8"), ("dst", "<u8"), ("n", "<i8"), ("item0", "<i8")])

        def table(pairs):
            recs, tot = [], 0
            for src, dst, n in pairs:
                recs.append((src, dst, n, tot))
                tot += n
            return torch.from_numpy(np.array(recs, dtype=dt).view(np.uint8).copy()).to(dev), len(recs), tot
        fw, bw, off = [], [], 0
        for w, bb, n in zip(weights, biases, self.sizes):
            fw.append((w.data_ptr(), self.W.data_ptr() + off * K * 4, n * K))
            fw.append((bb.data_ptr(), self.b.data_ptr() + off * 4, n))
            if self.sinks:
                bw.append((self.dW.data_ptr() + off * K * 4, grad_sink(w).data_ptr(), n * K))
                bw.append((self.db.data_ptr() + off * 4, grad_sink(bb).data_ptr(), n))
            off += n
        self.fwd_table, self.bwd_table = table(fw), (table(bw) if self.sinks else None)
        self.offs = [0]
        for n in self.sizes:
            self.offs.append(self.offs[-1] + n)
        self.tag, self.compute, self.wt = None, {}, {}

    def refresh(self):
        tag = _WEIGHT_EPOCH[0]
        if self.tag != tag:
            t, nd, tot = self.fwd_table
            _chk(_lib.load().cr_multi_seg(_ctx(self.W), _lib.ptr(t), nd, tot, 0), "cr_multi_seg")
            self.tag, self.compute, self.wt = tag, {}, {}

    def weight(self, dtype):
        """the stacked weight in the activation dtype (f32: the stack itself)"""
        if dtype not in self.compute:
            self.compute[dtype] = self.W if dtype == f32 else _act_cast(self.W, dtype)
        return self.compute[dtype]

    def weight_t(self, dtype):
        if dtype not in self.wt:
            _p = _Args()
            Wc = self.weight(dtype)
            wt = torch.empty((self.K, self.Op), dtype=dtype, device=Wc.device)
            _chk(_lib.load().cr_transpose2d(_ctx(Wc), _p(Wc), _p(wt), self.Op, self.K, int(dtype == f32)), "cr_transpose2d")
            self.wt[dtype] = wt
        return self.wt[dtype]


import collections as _collections
_CAT_PLANS = _collections.OrderedDict()      # least recently used plans are dropped beyond _CAT_PLANS_MAX
_CAT_PLANS_MAX = 64
_PLAN_KEEP = [None]     # a list while a graph capture is open (graphed.capture_guard): the plans the captured kernels point into


def _cat_plan(weights, biases):
    """plans are keyed by the parameters' addresses and shapes (not kept on the tensor objects: the whole-step graph runs
    the model on fresh leaf tensors over the same storage, and a plan must exist BEFORE a capture starts -- building one
    uploads its tables)"""
    sink_ptr = lambda t: 0 if grad_sink(t) is None else grad_sink(t).data_ptr()
    # (the gradient sinks are part of the key: a later model whose parameters land on a freed model's addresses has its own
    # flat gradient, and the plan's backward table holds raw sink pointers)
    key = tuple((w.data_ptr(), b.data_ptr(), tuple(w.shape), sink_ptr(w), sink_ptr(b)) for w, b in zip(weights, biases))
    sinks = all(grad_sink(t) is not None for t in list(weights) + list(biases))
    plan = _CAT_PLANS.get(key)
    if plan is None or plan.sinks != sinks:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.CrError("linear_cat: the stacked-predictor plan must be built before graph capture (run one eager step)")
        plan = _CAT_PLANS[key] = _CatPlan(weights, biases)
        while len(_CAT_PLANS) > _CAT_PLANS_MAX:
            _CAT_PLANS.popitem(last=False)
    else:
        _CAT_PLANS.move_to_end(key)
    if _PLAN_KEEP[0] is not None:
        _PLAN_KEEP[0].append(plan)          # the graph owner keeps the plan (its raw buffers) alive past an eviction
    return plan


class _LinearCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, plan, *params):
        plan.refresh()
        xc = x.contiguous()
        y = linear_fwd_raw(xc, plan.weight(xc.dtype), plan.b, out_f32=True)
        ctx.save_for_backward(xc)
        ctx.plan = plan
        return y

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        plan = ctx.plan
        dt = xc.dtype
        dy = _act_cast(dy, dt)
        dx = linear_bwd_data_raw(dy, plan.weight_t(dt)) if ctx.needs_input_grad[0] else None
        n = len(plan.sizes)
        if not any(ctx.needs_input_grad[2:]):
            return (dx, None) + (None,) * (2 * n)
        plan.db.zero_()
        linear_bwd_weight_raw(dy, xc, plan.dW, plan.db, accumulate=False)
        if plan.sinks:
            t, nd, tot = plan.bwd_table
            _chk(_lib.load().cr_multi_seg(_ctx(plan.dW), _lib.ptr(t), nd, tot, 1), "cr_multi_seg")
            return (dx, None) + (None,) * (2 * n)
        gw = [plan.dW[plan.offs[i]:plan.offs[i + 1]].clone() for i in range(n)]
        gb = [plan.db[plan.offs[i]:plan.offs[i + 1]].clone() for i in range(n)]
        return (dx, None) + tuple(gw) + tuple(gb)


def linear_cat(x, weights, biases):
    """[x @ w.T + b for w, b in zip(weights, biases)] as ONE GEMM: the predictor layers of a head share their input, so
    their weights are stacked (rows zero-padded to a multiple of 16 for the MFMA tiles) in persistent buffers refreshed by
    one launch per optimizer step; the output is the padded (R, O_pad) f32 matrix plus the column offset of each predictor.
    The stacked gradient goes back into the parameters' gradient sinks with one launch (or as per-parameter slices)."""
    weights, biases = list(weights), list(biases)
    plan = _cat_plan(weights, biases)
    return _LinearCat.apply(x, plan, *weights, *biases), list(plan.offs)


# --------------------------------------------------------------------------
# NMS
# --------------------------------------------------------------------------
def nms_grouped(boxes, counts, thresh):
    """boxes (G,maxn,4) f32 sorted by descending score per group; counts (G,) int32 -> keep (G,maxn) bool."""
    _p = _Args()
    _need_cuda(boxes, "boxes")
    G, maxn, _ = boxes.shape
    keep = torch.zeros((G, maxn), dtype=torch.uint8, device=boxes.device)
    if G == 0 or maxn == 0:
        return keep.bool()
    words = (maxn + 63) // 64
    ws = torch.empty((G * maxn * words,), dtype=torch.int64, device=boxes.device)
    lib = _lib.load()
    _chk(lib.cr_nms_grouped(_ctx(boxes), _p(boxes.contiguous()), _p(counts.to(torch.int32).contiguous()), G, maxn,
                            float(thresh), _p(ws), _p(keep)), "cr_nms_grouped")
    return keep.bool()


def det_select(logits, deltas, prop_boxes, objectness, img_hw, B, P, K, weights, scale_clamp, score_thresh, nms_thresh, detections,
               max_candidates=2048):
    """Test-time filter of the box head on B x P padded proposal slots, no host round trip (fast_rcnn.py:57-116): logits
    (B*P, >= K+1), deltas (B*P, 4K or 4), prop_boxes (B*P,4), objectness (B*P) or None, img_hw (B,2) float.
    -> boxes (B,D,4), scores (B,D), classes (B,D) int64, rows (B,D) int64 (proposal slot inside the image),
    scores_full (B,D,K), count (B,2) int32 = [detections, overflow flag]."""
    _p = _Args()
    _need_cuda(logits, "box head logits")
    lib = _lib.load()
    dev = logits.device
    # column slices of the predictor GEMM's output are taken as they are (row stride = its width): no copies
    rows_ok = lambda t: t.dtype == f32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]
    lg = logits.detach() if rows_ok(logits) else logits.detach().float().contiguous()
    dl = deltas.detach() if rows_ok(deltas) else deltas.detach().float().contiguous()
    ldl, ldd = lg.stride(0), dl.stride(0)
    pb = prop_boxes.detach().float().contiguous()
    nreg = 1 if dl.shape[1] < 4 * K else K
    S = torch.empty((B, P * K), dtype=f32, device=dev)
    ncand = torch.empty((B + 2,), dtype=torch.int32, device=dev)
    _chk(lib.cr_det_scores(_ctx(lg), _lib.P(lg.data_ptr()), ldl, _lib.P(dl.data_ptr()), ldd, _p(pb), _p(objectness), B, P, K, nreg, float(score_thresh),
                           _p(S), _p(ncand)), "cr_det_scores")
    Kc = min(int(max_candidates), P * K)
    val, idx = topk(S, Kc)
    boxes = torch.empty((B, Kc, 4), dtype=f32, device=dev)
    meta = torch.empty((3, B, Kc), dtype=torch.int32, device=dev)
    cls, row = meta[0], meta[1]
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    w4 = (_ct.c_float * 4)(*[float(w) for w in weights])
    val, idx = val.contiguous(), idx.contiguous()
    _chk(lib.cr_det_gather(_ctx(lg), _p(val), _p(idx), _lib.P(dl.data_ptr()), ldd, _p(pb), _p(img_hw.float().contiguous()), B, P, K, nreg, Kc,
                           w4, float(scale_clamp), _p(boxes), _p(cls), _p(row), _p(counts)), "cr_det_gather")
    words = (Kc + 63) // 64
    ws = torch.empty((B * Kc * words,), dtype=torch.int64, device=dev)
    keep = torch.empty((B, Kc), dtype=torch.uint8, device=dev)
    _chk(lib.cr_nms_grouped_cls(_ctx(lg), _p(boxes), _p(cls), _p(counts), B, Kc, float(nms_thresh), _p(ws), _p(keep)), "cr_nms_grouped_cls")
    D = int(detections)
    ob = torch.empty((B, D, 4), dtype=f32, device=dev)
    osc = torch.empty((B, D), dtype=f32, device=dev)
    oi = torch.empty((2, B, D), dtype=torch.int64, device=dev)
    ofull = torch.empty((B, D, K), dtype=f32, device=dev)
    ocnt = torch.empty((B, 2), dtype=torch.int32, device=dev)
    _chk(lib.cr_det_pick(_ctx(lg), _p(keep), _p(counts), _p(ncand), _p(val), _p(boxes), _p(cls), _p(row), _lib.P(lg.data_ptr()), ldl, B, P, K, Kc,
                         D, _p(ob), _p(osc), _p(oi[0]), _p(oi[1]), _p(ofull), _p(ocnt)), "cr_det_pick")
    return ob, osc, oi[0], oi[1], ofull, ocnt


# --------------------------------------------------------------------------
# static-shape RPN training glue (csrc/dense_train.hip)
# --------------------------------------------------------------------------
import ctypes as _ct


def _f4(vals):
    return (_ct.c_float * 4)(*[float(v) for v in vals])


def rpn_decode_select(anchors, deltas, idx, scores, weights, scale_clamp, img_hw, min_size):
    """anchors (A,4), deltas (B,A,4), idx (B,S) int64 (-1 = empty), scores (B,S), img_hw (B,2) ->
    boxes (B,S,4) clipped, nms_boxes (B,S,4) (zero where invalid), valid (B,S) bool."""
    _p = _Args()
    _need_cuda(deltas, "deltas")
    B, A = deltas.shape[0], anchors.shape[0]
    S = idx.shape[1]
    dev = deltas.device
    boxes = torch.empty((B, S, 4), dtype=f32, device=dev)
    nmsb = torch.empty((B, S, 4), dtype=f32, device=dev)
    valid = torch.empty((B, S), dtype=torch.uint8, device=dev)
    lib = _lib.load()
    _chk(lib.cr_rpn_decode_select(_ctx(deltas), _p(anchors.contiguous()), _p(deltas.contiguous()), _p(idx.contiguous()),
                                  _p(scores.contiguous()), B, A, S, _f4(weights), float(scale_clamp), _p(img_hw.contiguous()),
                                  float(min_size), _p(boxes), _p(nmsb), _p(valid)), "cr_rpn_decode_select")
    return boxes, nmsb, valid.bool()


def box_match(boxes, gt_boxes, gt_classes, want_best=False):
    """boxes (R,4) or (B,R,4); gt_boxes (B,G,4); gt_classes (B,G) int64 -> max_iou (B,R), argmax (B,R) int32,
    max_ioa (B,R), best (B,G) int64 (packed, see include/cr3dod.h) or None."""
    _p = _Args()
    _need_cuda(gt_boxes, "gt_boxes")
    B, G = gt_classes.shape
    per_image = boxes.dim() == 3
    R = boxes.shape[-2]
    dev = gt_boxes.device
    mi = torch.empty((B, R), dtype=f32, device=dev)
    am = torch.empty((B, R), dtype=torch.int32, device=dev)
    ma = torch.empty((B, R), dtype=f32, device=dev)
    best = torch.empty((B, G), dtype=torch.int64, device=dev) if want_best else None
    lib = _lib.load()
    _chk(lib.cr_box_match(_ctx(gt_boxes), _p(boxes.contiguous()), int(per_image), _p(gt_boxes.contiguous()),
                          _p(gt_classes.contiguous()), B, R, G, _p(mi), _p(am), _p(ma), _p(best)), "cr_box_match")
    return mi, am, ma, best


def rpn_label(anchors, gt_boxes, gt_classes, max_iou, best, expo, lo, hi, labels3, eps):
    _p = _Args()
    B, A = max_iou.shape
    G = gt_classes.shape[1]
    dev = max_iou.device
    labels_pre = torch.empty((B, A), dtype=torch.int8, device=dev)
    out = torch.empty((B, A), dtype=torch.int32, device=dev)
    miou = torch.empty((B, A), dtype=f32, device=dev)
    keys = torch.empty((2, B, A), dtype=f32, device=dev)
    lib = _lib.load()
    _chk(lib.cr_rpn_label(_ctx(max_iou), _p(anchors.contiguous()), _p(gt_boxes.contiguous()), _p(gt_classes.contiguous()),
                          _p(max_iou), _p(best), _p(expo.contiguous()), B, A, G, float(lo), float(hi),
                          (_ct.c_int * 3)(*[int(v) for v in labels3]), float(eps), _p(labels_pre), _p(out), _p(miou),
                          _p(keys)), "cr_rpn_label")
    return labels_pre, out, miou, keys


def rpn_scatter(out, pos_idx, pos_key, neg_idx, neg_key, n_s, ioa, ignore_thresh):
    """in place on out (B,A) int32."""
    _p = _Args()
    B, A = out.shape
    lib = _lib.load()
    _chk(lib.cr_rpn_scatter(_ctx(out), _p(pos_idx.contiguous()), _p(pos_key.contiguous()), pos_idx.shape[1],
                            _p(neg_idx.contiguous()), _p(neg_key.contiguous()), neg_idx.shape[1], int(n_s),
                            _p(ioa.contiguous()), float(ignore_thresh), B, A, _p(out)), "cr_rpn_scatter")
    return out


class _RPNLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, deltas, anchors, labels, midx, gt_boxes, weights):
        _p = _Args()
        B, A = logits.shape
        G = gt_boxes.shape[1]
        dev = logits.device
        nb = (A + 255) // 256
        ws = torch.empty((B * nb * 6,), dtype=f32, device=dev)
        sums = torch.empty((6,), dtype=f32, device=dev)
        dl = torch.empty((B, A), dtype=f32, device=dev)
        dd = torch.empty((B, A, 4), dtype=f32, device=dev)
        lib = _lib.load()
        _chk(lib.cr_rpn_loss(_ctx(logits), _p(logits.detach().float().contiguous()), _p(deltas.detach().float().contiguous()),
                             _p(anchors.contiguous()), _p(labels.contiguous()), _p(midx.contiguous()),
                             _p(gt_boxes.contiguous()), B, A, G, _f4(weights), _p(ws), _p(sums), _p(dl), _p(dd)),
             "cr_rpn_loss")
        ctx.save_for_backward(dl, dd)
        ctx.set_materialize_grads(False)       # outputs nobody differentiates arrive as None, not as zero fills
        ctx.mark_non_differentiable(sums)
        return sums[0], sums[1], sums

    @staticmethod
    def backward(ctx, g_cls, g_loc, _gs):
        dl, dd = ctx.saved_tensors
        return (None if g_cls is None else dl * g_cls), (None if g_loc is None else dd * g_loc), None, None, None, None, None


def rpn_loss(logits, deltas, anchors, labels, midx, gt_boxes, weights):
    """-> (loss_cls_sum, loss_loc_sum, sums6) unnormalised; labels (B,A) int32, midx (B,A) int32."""
    return _RPNLoss.apply(logits, deltas, anchors, labels, midx, gt_boxes, tuple(weights))


def roi_label(max_iou, argmax, max_ioa, valid, gt_classes, expo, K, thr, ignore_thresh, eps):
    _p = _Args()
    B, R = max_iou.shape
    dev = max_iou.device
    cls = torch.empty((B, R), dtype=torch.int64, device=dev)
    miou = torch.empty((B, R), dtype=f32, device=dev)
    keys = torch.empty((2, B, R), dtype=f32, device=dev)
    lib = _lib.load()
    _chk(lib.cr_roi_label(_ctx(max_iou), _p(max_iou), _p(argmax), _p(max_ioa), _p(valid.to(torch.uint8).contiguous()),
                          _p(gt_classes.contiguous()), _p(expo.contiguous()), B, R, gt_classes.shape[1], int(K), float(thr),
                          float(ignore_thresh), float(eps), _p(cls), _p(miou), _p(keys)), "cr_roi_label")
    return cls, miou, keys


def roi_compact(fg_idx, fg_key, bg_idx, bg_key, n_s, boxes, cls, argmax):
    """-> boxes (B,n_s,4), valid (B,n_s) bool, classes (B,n_s) int64, gt_idx (B,n_s) int64, counts (B,2) int32."""
    _p = _Args()
    B, R = cls.shape
    dev = cls.device
    ob = torch.empty((B, n_s, 4), dtype=f32, device=dev)
    ov = torch.empty((B, n_s), dtype=torch.uint8, device=dev)
    oc = torch.empty((B, n_s), dtype=torch.int64, device=dev)
    og = torch.empty((B, n_s), dtype=torch.int64, device=dev)
    counts = torch.empty((B, 2), dtype=torch.int32, device=dev)
    lib = _lib.load()
    _chk(lib.cr_roi_compact(_ctx(cls), _p(fg_idx.contiguous()), _p(fg_key.contiguous()), fg_idx.shape[1],
                            _p(bg_idx.contiguous()), _p(bg_key.contiguous()), bg_idx.shape[1], int(n_s), _p(boxes.contiguous()),
                            _p(cls), _p(argmax), B, R, _p(ob), _p(ov), _p(oc), _p(og), _p(counts)), "cr_roi_compact")
    return ob, ov.bool(), oc, og, counts


class _BoxLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, deltas, valid, cls, pboxes, gt_idx, gt_boxes, weights, scale_clamp):
        _p = _Args()
        B, S = valid.shape
        N, C = scores.shape
        K = C - 1
        dev = scores.device
        nb = (N + 3) // 4
        ws = torch.empty((nb * 3,), dtype=f32, device=dev)
        sums = torch.empty((3,), dtype=f32, device=dev)
        ds = torch.empty((N, C), dtype=f32, device=dev)
        dd = torch.empty((N, K * 4), dtype=f32, device=dev)
        pred = torch.empty((N, 4), dtype=f32, device=dev)
        lib = _lib.load()
        _chk(lib.cr_box_loss(_ctx(scores), _p(scores.detach().float().contiguous()), _p(deltas.detach().float().contiguous()),
                             _p(valid.to(torch.uint8).contiguous()), _p(cls.contiguous()), _p(pboxes.contiguous()),
                             _p(gt_idx.contiguous()), _p(gt_boxes.contiguous()), B, S, gt_boxes.shape[1], K, _f4(weights),
                             float(scale_clamp), _p(ws), _p(sums), _p(ds), _p(dd), _p(pred)), "cr_box_loss")
        ctx.save_for_backward(ds, dd)
        ctx.dt = (scores.dtype, deltas.dtype)
        ctx.set_materialize_grads(False)       # outputs nobody differentiates arrive as None, not as zero fills
        ctx.mark_non_differentiable(sums, pred)
        return sums[0], sums[1], sums, pred

    @staticmethod
    def backward(ctx, g_ce, g_l1, _gs, _gp):
        ds, dd = ctx.saved_tensors
        return (None if g_ce is None else (ds * g_ce).to(ctx.dt[0])), (None if g_l1 is None else (dd * g_l1).to(ctx.dt[1])), \
            None, None, None, None, None, None, None


def box_loss(scores, deltas, valid, cls, pboxes, gt_idx, gt_boxes, weights, scale_clamp):
    """-> (sum CE, sum L1, sums3 = [.., .., n_valid], pred (N,4)); valid/cls/gt_idx (B,S), pboxes (B,S,4)."""
    return _BoxLoss.apply(scores, deltas, valid, cls, pboxes, gt_idx, gt_boxes, tuple(weights), scale_clamp)


# --------------------------------------------------------------------------
# 3D head on the static-shape path: class gather + fused decode/loss + reductions
# --------------------------------------------------------------------------
CUBE_OFF = (0, 2, 3, 6, 15, 16, 20, 21, 24, 26, 27, 30)      # chunk starts of buf39 (x n), see include/cr3dod.h
Z_TYPES = {"direct": 0, "sigmoid": 1, "log": 2}           # MODEL.ROI_CUBE_HEAD.Z_TYPE values the kernels decode (roi_heads.py:2404-2410)


Z_TYPES["clusters"] = 3


def z_type_code(z_type):
    if z_type not in Z_TYPES:
        raise ValueError(f"Z_TYPE '{z_type}' is not built (built: {sorted(Z_TYPES)})")
    return Z_TYPES[z_type]


def z_config(z_type="direct", bins=1, z_scales=None, z_stats=None):
    """(code, bins, z_scales (K,bins) f32 or None, z_stats (K,bins,2) f32 or None): how a RoI's depth is read from the predictor
    output -- MODEL.ROI_CUBE_HEAD.Z_TYPE / CLUSTER_BINS with the head's priors_z_scales / priors_z_stats (roi_heads.py:2343-2436)"""
    code, bins = z_type_code(z_type), int(bins)
    if bins > 1 and z_scales is None:
        raise ValueError("CLUSTER_BINS > 1 needs priors_z_scales")
    if code == 3 and (bins <= 1 or z_stats is None):
        raise ValueError("Z_TYPE 'clusters' needs CLUSTER_BINS > 1 and priors_z_stats")
    f = lambda t: None if t is None else t.detach().float().contiguous()
    return (code, bins, f(z_scales) if bins > 1 else None, f(z_stats) if code == 3 else None)
CUBE_DIM = (2, 1, 3, 9, 1, 4, 1, 3, 2, 1, 3, 9)


def _chunks(buf, n):
    return [buf[o * n:(o + d) * n] for o, d in zip(CUBE_OFF, CUBE_DIM)]


class _CubeHeadLoss(torch.autograd.Function):
    """raw (n,13K) -> per-RoI losses (n,5), selected uncertainty (n); saves what the two backward kernels need."""

    @staticmethod
    def forward(ctx, raw, layout, K, cls, valid, gt_idx, kf, gt3d, gtpose, priors, meta, boxes, flags):
        _p = _Args()
        _need_cuda(raw, "cube head output")
        B, S = cls.shape
        n = B * kf
        dev = raw.device
        raw32 = raw.detach().float().contiguous()
        buf = torch.empty((39 * n,), dtype=f32, device=dev)
        validf = torch.empty((n,), dtype=torch.uint8, device=dev)
        clsc = torch.empty((n,), dtype=torch.int32, device=dev)
        lay = (_ct.c_int * 5)(*[int(v) for v in layout])
        lib = _lib.load()
        boxes = boxes.float().contiguous()
        _chk(lib.cr_cube_select(_ctx(raw), _p(raw32), raw32.shape[1], lay, int(K), _p(cls.contiguous()),
                                _p(valid.to(torch.uint8).contiguous()), _p(gt_idx.contiguous()), B, S, int(kf), gt3d.shape[1],
                                _p(gt3d.contiguous()), _p(gtpose.contiguous()), _p(priors), _p(meta.contiguous()), _p(buf),
                                _p(validf), _p(clsc), flags[4][0], flags[4][1], _p(flags[4][2]), _p(flags[4][3]), _p(boxes)),
             "cr_cube_select")
        ch = _chunks(buf, n)
        ins = ch[:5] + [boxes] + ch[5:]
        arr = (ctypes.c_void_p * 13)(*[t.data_ptr() for t in ins])
        losses = torch.empty((n, 5), dtype=f32, device=dev)
        dec = torch.empty((n, 17), dtype=f32, device=dev)
        _chk(lib.cr_cube_loss_fwd(_ctx(raw), ctypes.cast(arr, ctypes.c_void_p), n, *flags[:4], _p(losses), _p(dec)),
             "cr_cube_loss_fwd")
        ctx.keep = (raw32, buf, validf, clsc, boxes, tuple(layout), int(K), B, int(kf), flags, raw.dtype)
        ctx.set_materialize_grads(False)       # outputs nobody differentiates arrive as None, not as zero fills
        ctx.mark_non_differentiable(dec, buf, validf)
        return losses, ch[4].clone(), dec, buf, validf

    @staticmethod
    def backward(ctx, gl, g_usel, _gd, _gb, _gv):
        _p = _Args()
        raw32, buf, validf, clsc, boxes, layout, K, B, kf, flags, dt = ctx.keep
        n = B * kf
        dev = raw32.device
        ch = _chunks(buf, n)
        ins = ch[:5] + [boxes] + ch[5:]
        arr = (ctypes.c_void_p * 13)(*[t.data_ptr() for t in ins])
        g = torch.empty((16 * n,), dtype=f32, device=dev)
        g_dxy, g_zr, g_dr, g_Ra, g_u = g[:2 * n], g[2 * n:3 * n], g[3 * n:6 * n], g[6 * n:15 * n], g[15 * n:]
        lib = _lib.load()
        _chk(lib.cr_cube_loss_bwd(_ctx(raw32), ctypes.cast(arr, ctypes.c_void_p), n, *flags[:4], _p(gl.contiguous()),
                                  _p(g_dxy), _p(g_zr), _p(g_dr), _p(g_Ra), _p(g_u)), "cr_cube_loss_bwd")
        g_raw = torch.empty_like(raw32)
        lay = (_ct.c_int * 5)(*layout)
        _chk(lib.cr_cube_select_bwd(_ctx(raw32), _p(raw32), raw32.shape[1], lay, K, B, kf, _p(validf), _p(clsc), _p(g_dxy),
                                    _p(g_zr), _p(g_dr), _p(g_Ra), _p(g_u), _p(g_usel.contiguous()), _p(g_raw), flags[4][0], flags[4][1],
                                    _p(flags[4][2]), _p(flags[4][3]), _p(boxes)), "cr_cube_select_bwd")
        return (g_raw.to(dt),) + (None,) * 12


def cube_head_loss(raw, layout, K, cls, valid, gt_idx, kf, gt3d, gtpose, priors, meta, boxes, allocentric=True,
                   chamfer_pose=True, use_conf=True, joint=True, z_type="direct", z_cfg=None):
    """raw (n,13K) fused predictor output; cls/valid/gt_idx (B,S); gt3d (B,G,9); gtpose (B,G,3,3); priors (K,3) or None;
    meta (B,5); boxes (n,4).  -> losses (n,5), u_sel (n), dec (n,17), buf39, validf (n) uint8."""
    flags = (int(bool(allocentric)), int(bool(chamfer_pose)), int(bool(use_conf)), int(bool(joint)), z_cfg or z_config(z_type))
    return _CubeHeadLoss.apply(raw, tuple(layout), K, cls, valid, gt_idx, kf, gt3d, gtpose.reshape(gtpose.shape[0], -1, 9),
                               priors, meta, boxes, flags)


def cube_decode_infer(raw, layout, K, cls, img, boxes, meta6, priors, allocentric=True, z_type="direct", z_cfg=None):
    """inference decode of the 3D head (no autograd) -> (n,42), see cr_cube_decode_infer."""
    _p = _Args()
    _need_cuda(raw, "cube head output")
    n = raw.shape[0]
    out = torch.empty((n, 42), dtype=f32, device=raw.device)
    raw32 = raw.detach().float().contiguous()
    lay = (_ct.c_int * 5)(*[int(v) for v in layout])
    lib = _lib.load()
    zc = z_cfg or z_config(z_type)
    _chk(lib.cr_cube_decode_infer(_ctx(raw), _p(raw32), raw32.shape[1], lay, int(K), _p(cls.contiguous()),
                                  _p(img.to(torch.int32).contiguous()), _p(boxes.float().contiguous()), _p(meta6.contiguous()),
                                  _p(priors), n, int(bool(allocentric)), _p(out), zc[0], zc[1], _p(zc[2]), _p(zc[3])), "cr_cube_decode_infer")
    return out


class _CubeReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, L, u_sel, buf, dec, validf, inverse_z):
        _p = _Args()
        n = L.shape[0]
        dev = L.device
        out = torch.empty((16,), dtype=f32, device=dev)
        red, cnt, stats = out[:6], out[6:12], out[12:]
        lib = _lib.load()
        Lc = L.detach().contiguous()
        _chk(lib.cr_cube_reduce(_ctx(L), _p(Lc), _p(buf), _p(dec), _p(validf), n, int(inverse_z), _p(red), _p(cnt),
                                _p(stats)), "cr_cube_reduce")
        ctx.keep = (Lc, buf, validf, cnt, int(inverse_z))
        ctx.set_materialize_grads(False)       # outputs nobody differentiates arrive as None, not as zero fills
        ctx.mark_non_differentiable(stats)
        return red, stats

    @staticmethod
    def backward(ctx, gred, _gs):
        _p = _Args()
        Lc, buf, validf, cnt, inverse_z = ctx.keep
        n = Lc.shape[0]
        gL = torch.empty_like(Lc)
        gu = torch.empty((n,), dtype=f32, device=Lc.device)
        lib = _lib.load()
        _chk(lib.cr_cube_reduce_bwd(_ctx(Lc), _p(Lc), _p(buf), _p(validf), n, inverse_z, _p(cnt), _p(gred.contiguous()),
                                    _p(gL), _p(gu)), "cr_cube_reduce_bwd")
        return gL, gu, None, None, None, None


def cube_reduce(L, u_sel, buf, dec, validf, inverse_z=False):
    """-> red (6) = means of [dims, xy, z, pose, joint, uncert] over the valid finite entries, stats (4)."""
    return _CubeReduce.apply(L, u_sel, buf, dec, validf, inverse_z)


# --------------------------------------------------------------------------
# weakly supervised 3D head on the static-shape path: class gather + fused decode / losses / reductions
# --------------------------------------------------------------------------
WEAK_TERMS = ("iou", "pose", "normal", "z", "pseudo_gt_z", "dims_w", "dims_h", "dims_l")      # bit k of `terms`
WEAK_MAX_SLOTS = 256                                                                           # kf limit of k_weak_fwd / k_weak_bwd
_WEAK_IMG = {}


class _WeakCubeLoss(torch.autograd.Function):
    """raw (n,13K) -> red (9) = safely reduced, uncertainty-weighted terms + the mean uncertainty; see include/cr3dod.h
    (cr_weak_loss_fwd / _reduce / _bwd).  Five launches forward (select, terms, window medians, reduce), two backward."""

    @staticmethod
    def forward(ctx, raw, layout, K, cls, valid, gt_idx, kf, gt_boxes, gt3d, gtpose, prior_mean, prior_std, meta, table, normals,
                boxes, depth, terms, pgz_mode, weights, allocentric, zt):
        _p = _Args()
        _need_cuda(raw, "cube head output")
        B, S = cls.shape
        n, G = B * kf, gt3d.shape[1]
        dev = raw.device
        lib = _lib.load()
        raw32 = raw.detach().float().contiguous()
        buf = torch.empty((39 * n,), dtype=f32, device=dev)
        validf = torch.empty((n,), dtype=torch.uint8, device=dev)
        clsc = torch.empty((n,), dtype=torch.int32, device=dev)
        lay = (_ct.c_int * 5)(*[int(v) for v in layout])
        cls, gt_idx, gt_boxes, table = cls.contiguous(), gt_idx.contiguous(), gt_boxes.float().contiguous(), table.contiguous()
        boxes = boxes.float().contiguous()
        _chk(lib.cr_cube_select(_ctx(raw), _p(raw32), raw32.shape[1], lay, int(K), _p(cls), _p(valid.to(torch.uint8).contiguous()),
                                _p(gt_idx), B, S, int(kf), G, _p(gt3d.contiguous()), _p(gtpose.contiguous()), _p(prior_mean),
                                _p(meta.contiguous()), _p(buf), _p(validf), _p(clsc), zt[0], zt[1], _p(zt[2]), _p(zt[3]), _p(boxes)), "cr_cube_select")
        ch = _chunks(buf, n)
        ins = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in (ch[0], ch[1], ch[2], ch[3], ch[4], ch[6], ch[7], boxes)])
        out = torch.empty((n * (8 + 17 + 4 + 1 + 1) + 2 * B + 27,), dtype=f32, device=dev)
        o = 0
        def take(m):
            nonlocal o
            t = out[o:o + m]
            o += m
            return t
        Lraw, dec, pbox, ztgt, med, pimg = take(8 * n), take(17 * n), take(4 * n), take(n), take(n), take(2 * B)
        red, cnt, stats, aux = take(9), take(9), take(8), take(1)
        ibox = torch.empty((n, 4), dtype=torch.int32, device=dev)
        common = (_ctx(raw), ctypes.cast(ins, ctypes.c_void_p), _p(validf), _p(clsc), _p(gt_idx), _p(gt_boxes), _p(prior_std), _p(table),
                  _p(normals), B, int(kf), S, G, int(bool(allocentric)), int(terms))
        _chk(lib.cr_weak_loss_fwd(*common, _p(Lraw), _p(dec), _p(pbox), _p(ibox), _p(pimg)), "cr_weak_loss_fwd")
        H = W = 0
        if pgz_mode:
            depth = depth.float().contiguous()
            H, W = depth.shape[1], depth.shape[2]
        if pgz_mode == 1:
            img = _WEAK_IMG.get((B, kf, str(dev)))
            if img is None:
                img = _WEAK_IMG[(B, kf, str(dev))] = (torch.arange(n, device=dev) // kf).to(torch.int32)
            _chk(lib.cr_box_median(_ctx(raw), _p(depth), B, H, W, _p(ibox), _p(img), n, _p(med)), "cr_box_median")
        rin = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in (ch[4], ch[8], ch[9], ch[10])])
        wts = (_ct.c_float * 9)(*[float(w) for w in weights])
        _chk(lib.cr_weak_loss_reduce(_ctx(raw), ctypes.cast(rin, ctypes.c_void_p), _p(validf), _p(table), _p(depth if pgz_mode else None),
                                     H, W, _p(med if pgz_mode == 1 else None), _p(gt_boxes), _p(gt_idx), B, int(kf), S, G, int(terms),
                                     int(pgz_mode), wts, _p(Lraw), _p(dec), _p(pbox), _p(ibox), _p(pimg), _p(ztgt), _p(red), _p(cnt),
                                     _p(stats), _p(aux)), "cr_weak_loss_reduce")
        ctx.keep = (raw32, buf, validf, clsc, boxes, tuple(layout), int(K), B, int(kf), S, G, raw.dtype, cls, gt_idx, gt_boxes, prior_std,
                    table, normals, int(bool(allocentric)), int(terms), out, zt)
        dec2, pbox2 = dec.view(n, 17), pbox.view(n, 4)
        ctx.set_materialize_grads(False)       # outputs nobody differentiates arrive as None, not as zero fills
        ctx.mark_non_differentiable(stats, dec2, pbox2, validf)
        return red, stats, dec2, pbox2, validf

    @staticmethod
    def backward(ctx, gred, *_unused):
        _p = _Args()
        (raw32, buf, validf, clsc, boxes, layout, K, B, kf, S, G, dt, cls, gt_idx, gt_boxes, prior_std, table, normals, allocentric, terms,
         out, zt) = ctx.keep
        n = B * kf
        dev = raw32.device
        lib = _lib.load()
        ch = _chunks(buf, n)
        ins = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in (ch[0], ch[1], ch[2], ch[3], ch[4], ch[6], ch[7], boxes)])
        Lraw, dec, ztgt, pimg = out[:8 * n], out[8 * n:25 * n], out[29 * n:30 * n], out[31 * n:31 * n + 2 * B]
        tail = out[31 * n + 2 * B:]
        cnt, aux = tail[9:18], tail[26:27]
        g = torch.empty((17 * n,), dtype=f32, device=dev)
        g_dxy, g_zr, g_dr, g_Ra, g_u, zero = g[:2 * n], g[2 * n:3 * n], g[3 * n:6 * n], g[6 * n:15 * n], g[15 * n:16 * n], g[16 * n:]
        zero.zero_()
        _chk(lib.cr_weak_loss_bwd(_ctx(raw32), ctypes.cast(ins, ctypes.c_void_p), _p(validf), _p(clsc), _p(gt_idx), _p(gt_boxes),
                                  _p(prior_std), _p(table), _p(normals), B, kf, S, G, allocentric, terms, _p(gred.float().contiguous()),
                                  _p(cnt), _p(aux), _p(Lraw), _p(dec), _p(ztgt), _p(pimg), _p(g_dxy), _p(g_zr), _p(g_dr), _p(g_Ra),
                                  _p(g_u)), "cr_weak_loss_bwd")
        g_raw = torch.empty_like(raw32)
        lay = (_ct.c_int * 5)(*layout)
        _chk(lib.cr_cube_select_bwd(_ctx(raw32), _p(raw32), raw32.shape[1], lay, K, B, kf, _p(validf), _p(clsc), _p(g_dxy), _p(g_zr),
                                    _p(g_dr), _p(g_Ra), _p(g_u), _p(zero), _p(g_raw), zt[0], zt[1], _p(zt[2]), _p(zt[3]), _p(boxes)), "cr_cube_select_bwd")
        return (g_raw.to(dt),) + (None,) * 21


def weak_cube_loss(raw, layout, K, cls, valid, gt_idx, kf, gt_boxes, gt3d, gtpose, prior_mean, prior_std, meta, table, normals, boxes,
                   depth, terms, pgz_mode, weights, allocentric=True, z_type="direct", z_cfg=None):
    """Losses of the weakly supervised 3D head on the (B, kf) foreground slots (ROIHeads3DScore._forward_cube, training).
    raw (n,13K) fused predictor output; cls / valid / gt_idx (B,S); gt_boxes (B,G,4); gt3d (B,G,9); gtpose (B,G,3,3);
    prior_mean / prior_std (K,3) or None; meta (B,5) camera_meta(); table (B,20), normals (B,3) or None, depth (B,H,W) or
    None: see cr_weak_loss_fwd in include/cr3dod.h; terms: bit mask over WEAK_TERMS; pgz_mode 0 / 1 (window median) / 2 (depth
    under the centre); weights (9).  -> red (9), stats (8), dec (n,17), pbox (n,4), validf (n)."""
    if kf > WEAK_MAX_SLOTS:
        raise _lib.CrError(f"weak_cube_loss: {kf} foreground slots per image, the kernels take up to {WEAK_MAX_SLOTS}")
    return _WeakCubeLoss.apply(raw, tuple(layout), K, cls, valid, gt_idx, kf, gt_boxes, gt3d, gtpose.reshape(gtpose.shape[0], -1, 9),
                               prior_mean, prior_std, meta, table, normals, boxes, depth, terms, pgz_mode, tuple(weights), allocentric,
                               z_cfg or z_config(z_type))


# --------------------------------------------------------------------------
# optimizer
# --------------------------------------------------------------------------
class _RPNUnpack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, *ys):
        _p = _Args()
        L, B, C = len(ys), ys[0].shape[0], ys[0].shape[-1]
        ys = [y.contiguous() for y in ys]
        cells = [y.numel() // (B * C) for y in ys]
        atot, amax = sum(cells) * A, max(cells) * A
        dev = ys[0].device
        logits = torch.empty((B, atot), dtype=f32, device=dev)
        deltas = torch.empty((B, atot, 4), dtype=f32, device=dev)
        padded = torch.empty((B, L, amax), dtype=f32, device=dev)
        yp, ca = (_ct.c_void_p * L)(*[y.data_ptr() for y in ys]), (_ct.c_int * L)(*cells)
        _p.keep.extend(ys)
        _chk(_lib.load().cr_rpn_unpack(_ctx(ys[0]), yp, ca, L, B, A, C, _lib.ptr(logits), _lib.ptr(deltas), _lib.ptr(padded)),
             "cr_rpn_unpack")
        ctx.cfg = (A, C, B, cells, [tuple(y.shape) for y in ys])
        ctx.set_materialize_grads(False)       # outputs nobody differentiates arrive as None, not as zero fills
        ctx.mark_non_differentiable(padded)
        return logits, deltas, padded

    @staticmethod
    def backward(ctx, dlogits, ddeltas, _dpad):
        _p = _Args()
        A, C, B, cells, shapes = ctx.cfg
        L = len(cells)
        ref = dlogits if dlogits is not None else ddeltas
        dys = [torch.empty(s, dtype=f32, device=ref.device) for s in shapes]
        dl = None if dlogits is None else dlogits.contiguous()
        dd = None if ddeltas is None else ddeltas.contiguous()
        dp, ca = (_ct.c_void_p * L)(*[d.data_ptr() for d in dys]), (_ct.c_int * L)(*cells)
        _chk(_lib.load().cr_rpn_pack_grad(_ctx(ref), _p(dl), _p(dd), dp, ca, L, B, A, C), "cr_rpn_pack_grad")
        return (None,) + tuple(dys)


class _RPNUnpackStacked(torch.autograd.Function):
    """_RPNUnpack on ONE tensor Y (1,1,sum_l B H_l W_l,C) whose row blocks are the levels (conv_bias_act_group(stacked=True)
    followed by the 1x1 predictor convolution): same two kernels, level pointers = offsets into Y, one gradient tensor back"""

    @staticmethod
    def forward(ctx, A, cells, B, Y):
        _p = _Args()
        Y = Y.contiguous()
        L, C = len(cells), Y.shape[-1]
        assert Y.numel() == B * sum(cells) * C, "stacked RPN output does not match the level sizes"
        atot, amax = sum(cells) * A, max(cells) * A
        dev = Y.device
        logits = torch.empty((B, atot), dtype=f32, device=dev)
        deltas = torch.empty((B, atot, 4), dtype=f32, device=dev)
        padded = torch.empty((B, L, amax), dtype=f32, device=dev)
        offs = [0]
        for c in cells:
            offs.append(offs[-1] + B * c * C * 4)
        yp, ca = (_ct.c_void_p * L)(*[Y.data_ptr() + o for o in offs[:-1]]), (_ct.c_int * L)(*cells)
        _p.keep.append(Y)
        _chk(_lib.load().cr_rpn_unpack(_ctx(Y), yp, ca, L, B, A, C, _lib.ptr(logits), _lib.ptr(deltas), _lib.ptr(padded)),
             "cr_rpn_unpack")
        ctx.cfg = (A, C, B, tuple(cells), tuple(Y.shape), offs)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(padded)
        return logits, deltas, padded

    @staticmethod
    def backward(ctx, dlogits, ddeltas, _dpad):
        _p = _Args()
        A, C, B, cells, shape, offs = ctx.cfg
        L = len(cells)
        ref = dlogits if dlogits is not None else ddeltas
        dY = torch.empty(shape, dtype=f32, device=ref.device)
        dl = None if dlogits is None else dlogits.contiguous()
        dd = None if ddeltas is None else ddeltas.contiguous()
        dp, ca = (_ct.c_void_p * L)(*[dY.data_ptr() + o for o in offs[:-1]]), (_ct.c_int * L)(*cells)
        _chk(_lib.load().cr_rpn_pack_grad(_ctx(ref), _p(dl), _p(dd), dp, ca, L, B, A, C), "cr_rpn_pack_grad")
        return None, None, None, dY


def rpn_unpack_stacked(Y, cells, B, A):
    """rpn_unpack for the stacked head output: Y (1,1,B*sum(cells),C), cells = H_l*W_l per level"""
    return _RPNUnpackStacked.apply(int(A), tuple(int(c) for c in cells), int(B), Y)


def rpn_unpack(ys, A):
    """per-level RPN head outputs y_l (B,H,W,16) f32 [A logits | 4A deltas | pad] -> (logits (B,Atot), deltas (B,Atot,4),
    per-level logits padded with -inf (B,L,amax)) with ONE launch each way (was a slice copy per level and output, three cats,
    a fill and five copies; backward: a zero-fill, a slice copy and an add per level and output)"""
    return _RPNUnpack.apply(int(A), *ys)


def gt_pack(gt_instances, G, boxes, classes, boxes3D, poses):
    """fills the padded ground-truth tensors (B,G,...) of the static-shape training path from per-image Instances in one
    launch (include/cr3dod.h, cr_gt_pack); the outputs may be freshly allocated or the static buffers of a captured graph"""
    B = len(gt_instances)
    keep = []

    def dev(t, dt):
        t = t.to(device=boxes.device, dtype=dt).contiguous()
        keep.append(t)
        return t.data_ptr()
    PA, IA = _ct.c_void_p * B, _ct.c_int * B
    bp, cp, b3, pp, cnt = PA(), PA(), PA(), PA(), IA()
    for i, g in enumerate(gt_instances):
        n = len(g)
        cnt[i] = n
        bp[i] = dev(g.gt_boxes.tensor, f32) if n else None
        cp[i] = dev(g.gt_classes, torch.int64) if n else None
        has3 = n and g.has("gt_boxes3D")
        b3[i] = dev(g.gt_boxes3D, f32) if has3 else None
        pp[i] = dev(g.gt_poses, f32) if has3 else None
    _chk(_lib.load().cr_gt_pack(_ctx(boxes), bp, cp, b3, pp, cnt, B, int(G), _lib.ptr(boxes), _lib.ptr(classes), _lib.ptr(boxes3D),
                                _lib.ptr(poses)), "cr_gt_pack")


def topk(x, k):
    """row-wise top-k of a 2-D float32 tensor: (values sorted descending, int64 indices), ties -> lower index first;
    own radix-select + bitonic merge (csrc/topk.hip: two launches, no memset nodes).  Shapes outside the kernel's range
    (k > 2048, too many candidates) fall back to torch.topk."""
    _p = _Args()
    _need_cuda(x, "topk input")
    assert x.dim() == 2 and x.dtype == f32
    rows, n = x.shape
    lib = _lib.load()
    nb = lib.cr_topk_blocks(n, k) if k >= 1 else 0
    if k < 1 or k > 2048 or k > n or nb * k > 16384:
        if torch.cuda.is_current_stream_capturing():
            # torch.topk's multi-block path zeroes its counters with hipMemsetAsync = memset NODES in the captured graph,
            # the cause of the memory faults between back-to-back replays of the whole-step graph
            raise _lib.CrError(f"topk: rows of {n} values with k = {k} are outside the range of csrc/topk.hip (k <= 2048, "
                               f"chunks x k <= 16384) and the torch.topk fallback must not be captured into a HIP graph; "
                               f"run this configuration with CR_GRAPHS=dense or none")
        return x.topk(k, dim=1)
    xc = x.contiguous()
    ws = torch.empty((rows * nb * k,), dtype=torch.int64, device=x.device)
    vals = torch.empty((rows, k), dtype=f32, device=x.device)
    idx = torch.empty((rows, k), dtype=torch.int64, device=x.device)
    _chk(lib.cr_topk(_ctx(x), _p(xc), rows, n, k, _p(ws), _p(vals), _p(idx)), "cr_topk")
    return vals, idx


def loss_guard(vals, scale, red, total, recent, stabilize, tolerance, gamma, flag):
    """divergence guard of train_net.py:202-220 in one launch (see include/cr3dod.h); all tensors on the device, in place"""
    _p = _Args()
    _chk(_lib.load().cr_loss_guard(_ctx(vals), _p(vals), vals.numel(), float(scale), _p(red), _p(total), _p(recent),
                                   int(bool(stabilize)), float(tolerance), float(gamma), _p(flag)), "cr_loss_guard")


def step_counters(flag, explode, success):
    _p = _Args()
    _chk(_lib.load().cr_step_counters(_ctx(flag), _p(flag), _p(explode), _p(success)), "cr_step_counters")


def nonfinite_flag(flat_grad, flag):
    _p = _Args()
    lib = _lib.load()
    _chk(lib.cr_nonfinite_flag(_ctx(flat_grad), _p(flat_grad), flat_grad.numel(), _p(flag)), "cr_nonfinite_flag")


def sgd_step(p, g, m, lr, momentum, weight_decay, grad_scale=1.0, skip_flag=None, lr_scale_dev=None, nesterov=False):
    """lr_scale_dev: optional device float; the step uses lr * lr_scale_dev[0] (read on the device -> graph-capturable)"""
    _p = _Args()
    lib = _lib.load()
    fn = lib.cr_sgd_step_nesterov if nesterov else lib.cr_sgd_step
    _chk(fn(_ctx(p), _p(p), _p(g), _p(m), p.numel(), float(lr), _p(lr_scale_dev), float(momentum),
            float(weight_decay), float(grad_scale), _p(skip_flag)), "cr_sgd_step")


def grad_clip_value(g, clip_value, grad_scale=1.0):
    """SOLVER.CLIP_GRADIENTS, CLIP_TYPE 'value': g = clamp(g * grad_scale, -clip_value, clip_value) in place"""
    _p = _Args()
    _chk(_lib.load().cr_grad_clip_value(_ctx(g), _p(g), g.numel(), float(clip_value), float(grad_scale)), "cr_grad_clip_value")


def grad_clip_norm(g, starts, counts, max_norm, norm_type=2.0, grad_scale=1.0, partial=None):
    """CLIP_TYPE 'norm': every parameter (starts[i], counts[i] inside the flat gradient g) scaled on its own by
    min(1, max_norm / (||g_i * grad_scale|| + 1e-6)), grad_scale folded in; partial: (len(starts) * 16,) float scratch"""
    _p = _Args()
    n = int(starts.numel())
    if partial is None:
        partial = torch.empty(n * 16, dtype=torch.float32, device=g.device)
    _chk(_lib.load().cr_grad_clip_norm(_ctx(g), _p(g), _p(starts), _p(counts), n, float(max_norm), float(norm_type),
                                       float(grad_scale), _p(partial)), "cr_grad_clip_norm")


def adam_tick(step, skip_flag=None):
    """advance the device-side count of applied Adam updates unless the step is skipped (cr_adam_tick)"""
    _p = _Args()
    _chk(_lib.load().cr_adam_tick(_ctx(step), _p(step), _p(skip_flag)), "cr_adam_tick")


def adam_step(p, g, exp_avg, exp_avg_sq, max_exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, skip_flag=None,
              lr_scale_dev=None, decoupled=False):
    """torch.optim.Adam (decoupled=False) / AdamW (True) on one hyper-parameter segment of the flat buffers; max_exp_avg_sq:
    the amsgrad state or None; step: device float advanced by adam_tick"""
    _p = _Args()
    _chk(_lib.load().cr_adam_step(_ctx(p), _p(p), _p(g), _p(exp_avg), _p(exp_avg_sq), _p(max_exp_avg_sq), p.numel(), float(lr),
                                  _p(lr_scale_dev), float(beta1), float(beta2), float(eps), float(weight_decay), float(grad_scale),
                                  int(bool(decoupled)), _p(step), _p(skip_flag)), "cr_adam_step")


# --------------------------------------------------------------------------
# Depth-Anything-V2 forward ops (inference only: no autograd)
# --------------------------------------------------------------------------
def attention(qkv, B, N, H, D, scale):
    """qkv (B*N, 3*H*D) bf16 = output of the qkv linear -> (B*N, H*D) bf16 (cr_attention_fwd)."""
    _p = _Args()
    _need_cuda(qkv, "attention input")
    assert qkv.dtype == bf16 and qkv.is_contiguous() and qkv.shape == (B * N, 3 * H * D)
    out = torch.empty((B * N, H * D), dtype=bf16, device=qkv.device)
    _chk(_lib.load().cr_attention_fwd(_ctx(qkv), _p(qkv), _p(out), B, N, H, D, float(scale)), "cr_attention_fwd")
    return out


def layernorm(x, gamma, beta, eps):
    _p = _Args()
    _need_cuda(x, "layernorm input")
    assert x.dtype == bf16 and x.is_contiguous() and x.dim() == 2
    y = torch.empty_like(x)
    _chk(_lib.load().cr_layernorm(_ctx(x), _p(x), _p(gamma.detach().float().contiguous()),
                                  _p(beta.detach().float().contiguous()), _p(y), x.shape[0], x.shape[1], float(eps)),
         "cr_layernorm")
    return y


def gelu_(x):
    _p = _Args()
    _need_cuda(x, "gelu input")
    assert x.dtype == bf16 and x.is_contiguous()
    _chk(_lib.load().cr_gelu_inplace(_ctx(x), _p(x), x.numel()), "cr_gelu_inplace")
    return x


def scale_residual(x, y, gamma=None):
    """x + gamma * y on (M,C) bf16"""
    _p = _Args()
    _need_cuda(x, "residual input")
    assert x.dtype == bf16 and y.dtype == bf16 and x.is_contiguous() and y.is_contiguous() and x.shape == y.shape
    out = torch.empty_like(x)
    g = gamma.detach().float().contiguous() if gamma is not None else None
    _chk(_lib.load().cr_scale_residual(_ctx(x), _p(x), _p(y), _p(g), _p(out), x.shape[0], x.shape[1]), "cr_scale_residual")
    return out


def scale_residual_layernorm(x, y, ls, gamma, beta, eps):
    """(x + ls * y, LayerNorm(x + ls * y)) on (M,C) bf16 in one kernel"""
    _p = _Args()
    _need_cuda(x, "residual input")
    assert x.dtype == bf16 and y.dtype == bf16 and x.is_contiguous() and y.is_contiguous() and x.shape == y.shape
    xo, ho = torch.empty_like(x), torch.empty_like(x)
    l = ls.detach().float().contiguous() if ls is not None else None
    _chk(_lib.load().cr_scale_residual_layernorm(_ctx(x), _p(x), _p(y), _p(l), _p(gamma.detach().float().contiguous()),
                                                 _p(beta.detach().float().contiguous()), _p(xo), _p(ho), x.shape[0],
                                                 x.shape[1], float(eps)), "cr_scale_residual_layernorm")
    return xo, ho


def resize_bilinear_ac(x, size):
    """F.interpolate(..., mode='bilinear', align_corners=True) on NHWC bf16"""
    _p = _Args()
    _need_cuda(x, "resize input")
    assert x.dtype == bf16 and x.is_contiguous() and x.dim() == 4
    B, h, w, C = x.shape
    Ho, Wo = int(size[0]), int(size[1])
    y = torch.empty((B, Ho, Wo, C), dtype=bf16, device=x.device)
    _chk(_lib.load().cr_resize_bilinear_ac(_ctx(x), _p(x), _p(y), B, h, w, Ho, Wo, C), "cr_resize_bilinear_ac")
    return y
