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
