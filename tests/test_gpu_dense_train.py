"""GPU: the fused static-shape training-glue kernels (csrc/dense_train.hip, through the C-ABI) against the oracle's
tensor-op restatement (oracle/cpu_backend.py) on the same seeded inputs.  Integer outputs must agree exactly, IoUs to
float rounding (same operation order, contraction off), losses to 1e-5."""
import importlib

import pytest
import torch

from oracle import cpu_backend as O

pytestmark = pytest.mark.gpu
ops = importlib.import_module("3dod_amd.hipops")
d2 = importlib.import_module("3dod_amd.d2lite")
DEV = torch.device("cuda:0")


def _anchors():
    gen = d2.DefaultAnchorGenerator(sizes=[[32], [64], [128], [256], [512]], aspect_ratios=[[0.5, 1.0, 2.0]] * 5,
                                    strides=[4, 8, 16, 32, 64])
    lv = gen([(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)], torch.device("cpu"))
    return torch.cat([a.tensor for a in lv])


def _gt(B, G, seed, img=256.0):
    g = torch.Generator().manual_seed(seed)
    ctr = torch.rand(B, G, 2, generator=g) * img
    wh = torch.rand(B, G, 2, generator=g) * 100 + 8
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], -1).clamp(0, img)
    cls = torch.randint(0, 5, (B, G), generator=g)
    cls[0, G - 2:] = -2                      # padding
    cls[1, 1] = -1                           # an ignore region
    cls[B - 1, :] = -2                       # an image without objects
    if B > 2:
        cls[B - 1, 0] = -1                   # ... but with an ignore region
    return boxes, cls


@pytest.mark.parametrize("per_image", [False, True])
def test_box_match(per_image):
    anchors = _anchors()
    B, G = 3, 8
    gtb, gtc = _gt(B, G, 1)
    boxes = anchors if not per_image else (anchors[None] + torch.arange(B)[:, None, None] * 3.0)
    mi, am, ma, best = ops.box_match(boxes.to(DEV), gtb.to(DEV), gtc.to(DEV), want_best=True)
    rmi, ram, rma, rbest = O.box_match(boxes, gtb, gtc, want_best=True)
    assert torch.equal(mi.cpu(), rmi) and torch.equal(ma.cpu(), rma)
    has = (gtc >= 0).any(1)
    assert torch.equal(am.cpu()[has], ram[has])
    assert torch.equal(best.cpu(), rbest)
    mi2, am2, ma2, none = ops.box_match(boxes.to(DEV), gtb.to(DEV), gtc.to(DEV))
    assert none is None and torch.equal(mi2, mi) and torch.equal(am2, am) and torch.equal(ma2, ma)


def test_more_than_64_ground_truth_rows():
    """crowded images (Hypersim / SUN RGB-D with ignore regions) carry more than 64 ground-truth rows: the matcher and the
    anchor labeller stage them through LDS in chunks of 64 -- same results as the oracle for G = 150 (objects, ignore
    regions and padding spread over all three chunks; the forced arg-max anchors of rpn.py:75 included)."""
    anchors = _anchors()
    B, G = 2, 150
    g = torch.Generator().manual_seed(41)
    ctr = torch.rand(B, G, 2, generator=g) * 256
    wh = torch.rand(B, G, 2, generator=g) * 100 + 8
    gtb = torch.cat([ctr - wh / 2, ctr + wh / 2], -1).clamp(0, 256)
    gtc = torch.randint(0, 5, (B, G), generator=g)
    gtc[:, ::7] = -1                                 # ignore regions in every chunk
    gtc[0, 100:] = -2                                # padding: image 0 has 100 rows, image 1 all 150
    A = anchors.shape[0]
    mi, am, ma, best = ops.box_match(anchors.to(DEV), gtb.to(DEV), gtc.to(DEV), want_best=True)
    rmi, ram, rma, rbest = O.box_match(anchors, gtb, gtc, want_best=True)
    assert torch.equal(mi.cpu(), rmi) and torch.equal(ma.cpu(), rma) and torch.equal(am.cpu(), ram)
    assert torch.equal(best.cpu(), rbest)
    assert int(ram.max()) >= 128, "matches land in the third chunk too"
    expo = torch.empty(2, B, A).exponential_(1.0, generator=g)
    lab, out, miou, keys = ops.rpn_label(anchors.to(DEV), gtb.to(DEV), gtc.to(DEV), mi, best, expo.to(DEV), 0.3, 0.7,
                                         [0, -1, 1], 1e-4)
    rlab, rout, rmiou, rkeys = O.rpn_label(anchors, gtb, gtc, rmi, rbest, expo, 0.3, 0.7, [0, -1, 1], 1e-4)
    assert torch.equal(lab.cpu(), rlab) and torch.equal(out.cpu(), rout) and torch.equal(miou.cpu(), rmiou)
    assert torch.allclose(keys.cpu(), rkeys, rtol=1e-6, atol=0)
    assert int((rout == 1).sum()) > 64, "more forced anchors than one LDS chunk holds"


def test_rpn_decode_select():
    anchors = _anchors()
    A, B, S = anchors.shape[0], 2, 700
    g = torch.Generator().manual_seed(3)
    deltas = torch.randn(B, A, 4, generator=g) * 0.5
    deltas[0, 5, 2] = 50.0                       # clamped by scale_clamp
    deltas[1, 7, 0] = float("nan")               # non-finite box -> invalid, zeros
    idx = torch.randint(0, A, (B, S), generator=g)
    idx[0, :3] = torch.tensor([5, 7, 9]); idx[1, :3] = torch.tensor([5, 7, 9])
    idx[:, -50:] = -1                            # empty slots
    scores = torch.randn(B, S, generator=g)
    scores[:, -50:] = float("-inf")
    hw = torch.tensor([[256.0, 256.0], [200.0, 240.0]])
    args = ((1.0, 1.0, 2.0, 2.0), 4.135, hw, 2.0)
    b, nb, v = ops.rpn_decode_select(anchors.to(DEV), deltas.to(DEV), idx.to(DEV), scores.to(DEV), args[0], args[1],
                                     hw.to(DEV), args[3])
    rb, rnb, rv = O.rpn_decode_select(anchors, deltas, idx, scores, *args)
    assert torch.equal(v.cpu(), rv) and 0 < int(rv.sum()) < rv.numel()
    assert torch.allclose(b.cpu(), rb, rtol=1e-5, atol=1e-4) and torch.allclose(nb.cpu(), rnb, rtol=1e-5, atol=1e-4)


def test_rpn_label_scatter_and_loss():
    anchors = _anchors()
    A, B, G = anchors.shape[0], 3, 8
    gtb, gtc = _gt(B, G, 5)
    g = torch.Generator().manual_seed(6)
    expo = torch.empty(2, B, A).exponential_(1.0, generator=g)
    mi, am, ma, best = ops.box_match(anchors.to(DEV), gtb.to(DEV), gtc.to(DEV), want_best=True)
    lab, out, miou, keys = ops.rpn_label(anchors.to(DEV), gtb.to(DEV), gtc.to(DEV), mi, best, expo.to(DEV), 0.3, 0.7,
                                         [0, -1, 1], 1e-4)
    rmi, ram, rma, rbest = O.box_match(anchors, gtb, gtc, want_best=True)
    rlab, rout, rmiou, rkeys = O.rpn_label(anchors, gtb, gtc, rmi, rbest, expo, 0.3, 0.7, [0, -1, 1], 1e-4)
    assert torch.equal(lab.cpu(), rlab) and torch.equal(out.cpu(), rout) and torch.equal(miou.cpu(), rmiou)
    assert torch.allclose(keys.cpu(), rkeys, rtol=1e-6, atol=0)
    assert int((rlab == 1).sum()) > 0 and int((rout == 1).sum()) > 0
    # sampling picks from the (identical) keys, scatter on both sides
    n_s, kp = 64, 32
    pkey, pidx = rkeys[0].topk(kp, dim=1)
    nkey, nidx = rkeys[1].topk(n_s, dim=1)
    o_gpu = ops.rpn_scatter(out.clone(), pidx.to(DEV), pkey.to(DEV), nidx.to(DEV), nkey.to(DEV), n_s, ma, 0.5)
    o_ref = O.rpn_scatter(rout.clone(), pidx, pkey, nidx, nkey, n_s, rma, 0.5)
    assert torch.equal(o_gpu.cpu(), o_ref)
    assert int((o_ref == 0).sum()) > 0 and int((o_ref == 1).sum()) > 0
    # losses and gradients
    logits = torch.randn(B, A, generator=g)
    deltas = torch.randn(B, A, 4, generator=g) * 0.2
    w = (1.0, 1.0, 1.0, 1.0)
    lg, dg = logits.to(DEV).requires_grad_(), deltas.to(DEV).requires_grad_()
    lc, ll, sums = ops.rpn_loss(lg, dg, anchors.to(DEV), o_gpu, am, gtb.to(DEV), w)
    (lc * 0.7 + ll * 1.3).backward()
    lr_, dr_ = logits.clone().requires_grad_(), deltas.clone().requires_grad_()
    rc, rl, rsums = O.rpn_loss(lr_, dr_, anchors, o_ref, ram, gtb, w)
    (rc * 0.7 + rl * 1.3).backward()
    assert torch.allclose(sums.cpu(), rsums, rtol=1e-5, atol=1e-5)
    assert abs(float(lc) - float(rc)) < 1e-5 * max(1, abs(float(rc))) and abs(float(ll) - float(rl)) < 1e-5 * max(1, abs(float(rl)))
    assert torch.allclose(lg.grad.cpu(), lr_.grad, rtol=1e-5, atol=1e-6)
    assert torch.allclose(dg.grad.cpu(), dr_.grad, rtol=1e-5, atol=1e-6)


def test_roi_label_compact_and_box_loss():
    B, G, R, K = 3, 8, 600, 5
    gtb, gtc = _gt(B, G, 11)
    g = torch.Generator().manual_seed(12)
    ctr = torch.rand(B, R, 2, generator=g) * 256
    wh = torch.rand(B, R, 2, generator=g) * 90 + 6
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], -1).clamp(0, 256)
    boxes[:, :G] = gtb                                   # some exact hits -> foreground
    boxes[:, G:3 * G] = gtb.repeat(1, 2, 1) + torch.randn(B, 2 * G, 4, generator=g) * 3
    valid = torch.rand(B, R, generator=g) > 0.1
    valid[B - 1] = torch.rand(R, generator=g) > 0.9      # fewer valid proposals than sample slots -> empty slots
    expo = torch.empty(2, B, R).exponential_(1.0, generator=g)
    mi, am, ma, _ = ops.box_match(boxes.to(DEV), gtb.to(DEV), gtc.to(DEV))
    rmi, ram, rma, _ = O.box_match(boxes, gtb, gtc)
    cls, miou, keys = ops.roi_label(mi, am, ma, valid.to(DEV), gtc.to(DEV), expo.to(DEV), K, 0.5, 0.5, 1e-4)
    rcls, rmiou, rkeys = O.roi_label(rmi, ram, rma, valid, gtc, expo, K, 0.5, 0.5, 1e-4)
    assert torch.equal(cls.cpu(), rcls) and torch.equal(miou.cpu(), rmiou)
    assert torch.allclose(keys.cpu(), rkeys, rtol=1e-6, atol=0)
    assert int(((rcls >= 0) & (rcls < K)).sum()) > 0 and int((rcls == -1).sum()) > 0 and int((rcls == K).sum()) > 0
    n_s, kf = 128, 32
    fkey, fidx = rkeys[0].topk(kf, dim=1)
    bkey, bidx = rkeys[1].topk(n_s, dim=1)
    outs = ops.roi_compact(fidx.to(DEV), fkey.to(DEV), bidx.to(DEV), bkey.to(DEV), n_s, boxes.to(DEV), cls, am)
    refs = O.roi_compact(fidx, fkey, bidx, bkey, n_s, boxes, rcls, ram)
    for a, b in zip(outs, refs):
        assert torch.equal(a.cpu(), b), (a.shape, a.dtype)
    s_boxes, s_valid, s_cls, s_gt, counts = refs
    assert int(counts[:, 0].sum()) > 0 and int(counts[:, 1].sum()) > 0 and not bool(s_valid.all())
    # Fast R-CNN losses on that sample
    N = B * n_s
    scores = torch.randn(N, K + 1, generator=g)
    deltas = torch.randn(N, K * 4, generator=g) * 0.3
    w, clamp = (10.0, 10.0, 5.0, 5.0), 4.135
    sg, dg = scores.to(DEV).requires_grad_(), deltas.to(DEV).requires_grad_()
    ce, l1, sums, pred = ops.box_loss(sg, dg, outs[1], outs[2], outs[0], outs[3], gtb.to(DEV), w, clamp)
    (ce * 0.3 + l1 * 1.7).backward()
    sr, dr = scores.clone().requires_grad_(), deltas.clone().requires_grad_()
    rce, rl1, rsums, rpred = O.box_loss(sr, dr, s_valid, s_cls, s_boxes, s_gt, gtb, w, clamp)
    (rce * 0.3 + rl1 * 1.7).backward()
    assert torch.allclose(sums.cpu(), rsums, rtol=1e-5, atol=1e-4)
    assert torch.allclose(pred.cpu(), rpred, rtol=1e-5, atol=1e-3)
    assert torch.allclose(sg.grad.cpu(), sr.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(dg.grad.cpu(), dr.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("inverse_z", [False, True])
def test_cube_head_loss_and_reduce(inverse_z):
    """class gather + 6D rotation + clip + fused decode/loss + safely_reduce_losses, forward and backward, against the
    oracle's tensor-op restatement (autograd through rotation_6d_to_matrix / indexing / masked means)."""
    B, S, kf, G, K = 3, 20, 6, 5, 7
    n = B * kf
    g = torch.Generator().manual_seed(21)
    raw = torch.randn(n, 13 * K, generator=g) * 0.3
    raw[:, 11 * K:12 * K] += 3.0                              # depths around 3 (x virtual_to_real)
    raw[:, 12 * K:] = torch.randn(n, K, generator=g) * 0.5 + 0.2   # some uncertainties below the 0.01 clip
    layout = (0, 2 * K, 5 * K, 11 * K, 12 * K)
    cls = torch.randint(0, K, (B, S), generator=g)
    cls[0, 1] = K                                             # background in a foreground slot -> invalid
    cls[1, 2] = -1
    valid = torch.rand(B, S, generator=g) > 0.2
    gt_idx = torch.randint(0, G, (B, S), generator=g)
    gt3d = torch.cat([torch.rand(B, G, 2, generator=g) * 400 + 50, torch.rand(B, G, 1, generator=g) * 8 + 2,
                      torch.rand(B, G, 3, generator=g) * 2 + 0.3, torch.rand(B, G, 3, generator=g)], -1)
    util = importlib.import_module("3dod_amd.cubercnn.util.math_util")
    gtpose = util.rotation_6d_to_matrix(torch.randn(B * G, 6, generator=g)).view(B, G, 3, 3)
    priors = torch.rand(K, 3, generator=g) + 0.5
    meta = torch.tensor([[500.0, 510, 256, 250, 1.0], [450, 450, 240, 260, 1.2], [520, 500, 250, 255, 0.9]])
    ctr = torch.rand(n, 2, generator=g) * 300 + 100
    wh = torch.rand(n, 2, generator=g) * 120 + 20
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    flags = dict(allocentric=True, chamfer_pose=True, use_conf=True, joint=True)
    wts = torch.tensor([1.0, 0.7, 1.3, 0.5, 2.0, 0.9])

    rg = raw.to(DEV).requires_grad_()
    L, u, dec, buf, vf = ops.cube_head_loss(rg, layout, K, cls.to(DEV), valid.to(DEV), gt_idx.to(DEV), kf, gt3d.to(DEV),
                                            gtpose.to(DEV), priors.to(DEV), meta.to(DEV), boxes.to(DEV), **flags)
    red, stats = ops.cube_reduce(L, u, buf, dec, vf, inverse_z=inverse_z)
    (red * wts.to(DEV)).sum().backward()

    rr = raw.clone().requires_grad_()
    rL, ru, rdec, rbuf, rvf = O.cube_head_loss(rr, layout, K, cls, valid, gt_idx, kf, gt3d, gtpose, priors, meta, boxes, **flags)
    rred, rstats = O.cube_reduce(rL, ru, rbuf, rdec, rvf, inverse_z=inverse_z)
    (rred * wts).sum().backward()

    assert torch.equal(vf.cpu(), rvf) and 0 < int(rvf.sum()) < n
    assert torch.allclose(buf.cpu(), rbuf, rtol=1e-5, atol=1e-5)
    assert torch.allclose(u.cpu(), ru.detach(), rtol=1e-6, atol=1e-6)
    m = rvf.bool()
    assert torch.allclose(L.detach().cpu()[m], rL.detach()[m], rtol=2e-4, atol=2e-4)
    assert torch.allclose(dec.cpu()[m], rdec[m], rtol=2e-4, atol=2e-3)
    assert torch.allclose(red.detach().cpu(), rred.detach(), rtol=2e-4, atol=2e-4)
    assert torch.allclose(stats.cpu(), rstats, rtol=2e-4, atol=2e-4)
    ga, gb = rg.grad.cpu(), rr.grad
    assert float((ga - gb).abs().max()) <= 2e-3 * float(gb.abs().max()) + 1e-6, float((ga - gb).abs().max())
    assert float(gb.abs().max()) > 0


def test_rpn_unpack_matches_slices_forward_and_backward():
    """cr_rpn_unpack / cr_rpn_pack_grad against the slice / reshape / cat formulation (oracle/cpu_backend.rpn_unpack =
    detectron2's RPN.forward layout [third-party], rpn.py:153-170): bit-exact forward, exact gradients"""
    from oracle import cpu_backend as O
    g = torch.Generator().manual_seed(4)
    A, B = 3, 2
    shapes = [(B, 16, 24, 16), (B, 8, 12, 16), (B, 4, 6, 16), (B, 2, 3, 16), (B, 1, 2, 16)]
    ys_c = [torch.randn(s, generator=g).requires_grad_(True) for s in shapes]
    ys_d = [y.detach().to(DEV).requires_grad_(True) for y in ys_c]
    lo, do, po = O.rpn_unpack(ys_c, A)
    lg, dg, pg_ = ops.rpn_unpack(ys_d, A)
    assert torch.equal(lg.cpu(), lo) and torch.equal(dg.cpu(), do) and torch.equal(pg_.cpu(), po)
    wl, wd = torch.randn(lo.shape, generator=g), torch.randn(do.shape, generator=g)
    ((lo * wl).sum() + (do * wd).sum()).backward()
    ((lg * wl.to(DEV)).sum() + (dg * wd.to(DEV)).sum()).backward()
    for a, b in zip(ys_c, ys_d):
        assert torch.equal(a.grad, b.grad.cpu())
    # only one of the two outputs used: the other gradient is None -> zeros
    ys_e = [y.detach().to(DEV).requires_grad_(True) for y in ys_c]
    l2, d2, _ = ops.rpn_unpack(ys_e, A)
    l2.sum().backward()
    assert all(float(y.grad[..., A:].abs().max()) == 0 and float(y.grad[..., :A].min()) == 1 for y in ys_e)
