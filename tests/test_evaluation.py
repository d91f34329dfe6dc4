"""CPU: the AP2D / AP3D scoring core (3dod_amd/cubercnn/evaluation) on a closed loop with hand-computed answers
(pycocotools / the reference's evaluator cannot be imported: parity unpinned w.r.t. them; the protocol constants are
the reference's, omni3d_evaluation.py:1020-1088)."""
import importlib

import numpy as np

from oracle import iou3d as O
from tests.test_iou3d import box

ev = importlib.import_module("3dod_amd.cubercnn.evaluation")


def oracle_iou3d(d, g):
    return O.box3d_overlap(np.asarray(d), np.asarray(g))[1]


def rec(img, cat, idx, center, dims, score=None, depth=None, ignore=0):
    c = box(center, dims)
    x, y = center[0] * 50 + 200, center[1] * 50 + 200
    r = {"image_id": img, "category_id": cat, "id": idx, "bbox": [x, y, dims[0] * 50, dims[1] * 50],
         "area": dims[0] * dims[1] * 2500, "bbox3D": c.tolist(), "depth": center[2] if depth is None else depth}
    if score is None:
        r.update(ignore2D=ignore, ignore3D=ignore)
    else:
        r["score"] = score
    return r


def test_params_are_the_references():
    p2, p3 = ev.Omni3DParams("2D"), ev.Omni3DParams("3D")
    assert np.allclose(p2.iouThrs, np.arange(10) * 0.05 + 0.5) and np.allclose(p3.iouThrs, np.arange(10) * 0.05 + 0.05)
    assert len(p3.recThrs) == 101 and p3.maxDets == [1, 10, 100]
    assert p3.areaRng == [[0, 1e5], [0, 10], [10, 35], [35, 1e5]] and p3.areaRngLbl == ["all", "near", "medium", "far"]


def test_hand_computed_ap():
    """TP, FP, TP in score order over 2 ground-truth objects: precision envelope [1, 2/3, 2/3], recall [.5, .5, 1] ->
    AP = (51 * 1 + 50 * 2/3) / 101 at every IoU threshold (the matches have IoU 1)."""
    gts = [rec(1, 0, 1, [0, 0, 5], [1, 1, 1]), rec(2, 0, 2, [1, 0, 6], [1, 2, 1])]
    dts = [rec(1, 0, 1, [0, 0, 5], [1, 1, 1], score=0.9), rec(1, 0, 2, [3, 3, 5], [1, 1, 1], score=0.8),
           rec(2, 0, 3, [1, 0, 6], [1, 2, 1], score=0.7)]
    want = (51 * 1.0 + 50 * 2.0 / 3.0) / 101
    for mode in ("2D", "3D"):
        e = ev.Omni3Deval(gts, dts, mode, iou3d_fn=oracle_iou3d).evaluate().accumulate()
        s = e.summarize()
        assert abs(s[0] - want) < 1e-9 and abs(s[1] - want) < 1e-9, (mode, s)
        assert abs(s[9] - 1.0) < 1e-12 and abs(s[7] - 1.0) < 1e-12            # AR@100 = 1; AR@1: the top detection of each image is a TP
    # depth ranges (3D): both objects are "near" (< 10 m) -> AP_near = AP, no medium / far ground truth -> -1
    assert abs(s[4] - want) < 1e-9 and s[5] == -1 and s[6] == -1


def test_iou_thresholds_ignore_and_ranges():
    # one object, detection shifted by half its width: IoU3D = 1/3 -> TP up to the 0.30 threshold, FP above
    gts = [rec(1, 0, 1, [0, 0, 20], [2, 2, 2])]
    dts = [rec(1, 0, 1, [1, 0, 20], [2, 2, 2], score=0.5)]
    e = ev.Omni3Deval(gts, dts, "3D", iou3d_fn=oracle_iou3d).evaluate().accumulate()
    s = e.summarize()
    n_tp = int((e.params.iouThrs <= 1 / 3 + 1e-12).sum())                     # 0.05 ... 0.30 -> 6 thresholds
    assert n_tp == 6 and abs(s[0] - n_tp / 10) < 1e-9
    assert abs(s[1] - 1.0) < 1e-9 and abs(s[2] - 1.0) < 1e-9 and abs(s[3] - 0.0) < 1e-9      # @0.15, @0.25, @0.50
    assert s[4] == -1 and abs(s[5] - n_tp / 10) < 1e-9 and s[6] == -1          # the object is "medium" (10-35 m)
    # an ignored ground truth absorbs its detection: neither TP nor FP; nothing left to evaluate -> -1
    gts[0]["ignore3D"] = 1
    s = ev.Omni3Deval(gts, dts, "3D", iou3d_fn=oracle_iou3d).evaluate().accumulate().summarize()
    assert s[0] == -1
    # ... and next to a real object: up to the 0.30 threshold the first detection is absorbed by the ignored object
    # (AP = 1); above it, it is an unmatched false positive ranked before the true positive (precision envelope 1/2)
    gts.append(rec(1, 0, 2, [8, 0, 20], [2, 2, 2]))
    dts.append(rec(1, 0, 2, [8, 0, 20], [2, 2, 2], score=0.4))
    s = ev.Omni3Deval(gts, dts, "3D", iou3d_fn=oracle_iou3d).evaluate().accumulate().summarize()
    assert abs(s[0] - (6 * 1.0 + 4 * 0.5) / 10) < 1e-9


def test_max_dets_and_categories():
    gts = [rec(1, c, c + 1, [3 * c, 0, 5], [1, 1, 1]) for c in range(3)]
    dts = [rec(1, c, c + 1, [3 * c, 0, 5], [1, 1, 1], score=0.9 - 0.1 * c) for c in range(2)]   # category 2 is missed
    s = ev.Omni3Deval(gts, dts, "3D", iou3d_fn=oracle_iou3d).evaluate().accumulate().summarize()
    assert abs(s[0] - 2 / 3) < 1e-9 and abs(s[9] - 2 / 3) < 1e-9               # mean over categories: 1, 1, 0
    assert np.allclose(ev.iou_xywh([[0, 0, 2, 2]], [[1, 0, 2, 2], [0, 0, 2, 2], [5, 5, 1, 1]]), [[1 / 3, 1.0, 0.0]])
