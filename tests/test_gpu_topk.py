"""cr_topk (csrc/topk.hip) against torch.topk on the CPU: values bit-exact, indices exact where values are distinct, ties
resolved to the lower index, for the four shapes of the train step (RPN anchor sampling 8 x 523776 / 256, pre-NMS
20 x 196608 / 2000, post-NMS 4 x 10000 / 1000, RoI sampling 8 x 1032 / 512), rows with -inf padding, heavy ties, NaN, and
the edge cases k = n and k = 1.  Replaces torch.topk behind subsample_labels (rpn.py:275-328 of the reference) and
detectron2's find_top_rpn_proposals."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu
ops = importlib.import_module("3dod_amd.hipops")
DEV = "cuda:0"


def check(x, k):
    v, i = ops.topk(x.to(DEV), k)
    torch.cuda.synchronize()
    v, i = v.cpu(), i.cpu()
    rv, _ = x.topk(k, dim=1)
    assert torch.equal(torch.nan_to_num(v, nan=7e37), torch.nan_to_num(rv, nan=7e37)), "values (sorted descending) must match torch.topk bit for bit"
    got = torch.gather(x, 1, i)
    assert torch.equal(torch.nan_to_num(got, nan=7e37), torch.nan_to_num(v, nan=7e37)), "indices must point at the returned values"
    for r in range(x.shape[0]):                      # no index twice; ties: ascending index inside a run of equal values
        assert len(set(i[r].tolist())) == k
        same = v[r, 1:] == v[r, :-1]
        assert (i[r, 1:][same] > i[r, :-1][same]).all()
        # lowest-index rule at the cut: every element equal to the k-th value and not chosen has a larger index
        kth = v[r, -1]
        if kth == kth:
            chosen = set(i[r][v[r] == kth].tolist())
            rest = [j for j in (x[r] == kth).nonzero().flatten().tolist() if j not in chosen]
            if rest and chosen:
                assert min(rest) > max(chosen)


@pytest.mark.parametrize("rows,n,k", [(8, 523776, 256), (20, 196608, 2000), (4, 10000, 1000), (8, 1032, 512),
                                      (3, 5, 5), (2, 40000, 1), (5, 32769, 2048)])
def test_topk_matches_torch(rows, n, k):
    g = torch.Generator().manual_seed(rows * 131 + k)
    check(torch.randn(rows, n, generator=g), k)


def test_topk_padding_ties_nan():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(6, 70000, generator=g)
    x[0, 100:] = float("-inf")                       # fewer finite values than k
    x[1] = torch.randint(0, 4, (70000,), generator=g).float()          # heavy ties
    x[2] = 0.0
    x[3, ::7] = float("nan")
    x[4] = torch.rand(70000, generator=g) * 1e-3 + 0.5                 # one exponent: everything in one top-level bin
    x[5] = -torch.rand(70000, generator=g)
    check(x, 300)
    check(x, 2048)


def test_topk_negative_nan_sorts_first_like_torch():
    """torch.topk treats EVERY NaN as the largest value; a NaN with the sign bit set (0xffc00000, what 0 * -inf or
    -(0/0) produce) must not sort as the smallest key"""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 5000, generator=g)
    neg_nan = torch.tensor([0xffc00000 - (1 << 32)], dtype=torch.int32).view(torch.float32)[0]
    x[0, 17] = neg_nan
    x[1, ::9] = neg_nan
    x[1, 1::9] = float("nan")
    v, i = ops.topk(x.to(DEV), 40)
    v, i = v.cpu(), i.cpu()
    assert bool(torch.isnan(v[0, 0])) and int(i[0, 0]) == 17 and not bool(torch.isnan(v[0, 1:]).any())
    assert bool(torch.isnan(v[1]).all())
    rv, _ = x.topk(40, dim=1)
    assert torch.equal(torch.isnan(v), torch.isnan(rv))
    assert torch.equal(torch.nan_to_num(v, nan=0.0), torch.nan_to_num(rv, nan=0.0))


def test_topk_out_of_range_refuses_capture():
    """shapes outside the kernel's range fall back to torch.topk in eager mode, but NOT inside a stream capture: torch.topk's
    multi-block path would put hipMemsetAsync nodes into the graph (memory faults between replays)"""
    lib = importlib.import_module("3dod_amd._lib")
    x = torch.randn(2, 20000, device=DEV)
    v, i = ops.topk(x, 3000)                        # eager: allowed (k > 2048 -> torch.topk)
    assert torch.equal(v, x.topk(3000, dim=1)[0])
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.topk(x, 256)                            # lazy initialisation outside the capture
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    raised = False
    try:
        with torch.cuda.graph(g):
            ops.topk(x, 256)                        # in range: capturable
            try:
                ops.topk(x, 3000)
            except lib.CrError as e:
                raised = "must not be captured" in str(e)
    finally:
        torch.cuda.synchronize()
    assert raised
    g.replay()                                      # the capture itself stayed valid
    torch.cuda.synchronize()
