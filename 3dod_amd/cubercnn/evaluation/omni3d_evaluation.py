"""AP2D / AP3D evaluation -- the scoring core of cubercnn/evaluation/omni3d_evaluation.py of the reference
(Omni3DParams :1020-1088, Omni3Deval.computeIoU :1360-1432, evaluateImg :1434-1553, accumulate :1173-1315,
summarize :1555-1700; the protocol is pycocotools' COCOeval [third-party], restated).

Scope: in-memory ground truth / detection records -> the 13 summary numbers of each mode.  3D IoU goes
through the exact-IoU kernel (geometry.box3d_overlap = cr_box3d_overlap) instead of pytorch3d.  Dataset / json plumbing
(Omni3DEvaluator, inference_on_dataset, result tables) belongs to the data path (N2) and is not built.

Record fields (as written by instances_to_coco_json, :971-1014, and the Omni3D json): image_id, category_id, id,
bbox [x,y,w,h], area, bbox3D (8,3) corners, depth; detections add score; ground truth adds ignore2D / ignore3D."""
import collections

import numpy as np


class Omni3DParams:
    def __init__(self, mode="2D"):
        if mode == "2D":
            self.iouThrs = np.linspace(0.5, 0.95, 10)
            self.areaRng = [[0, 1e5 ** 2], [0, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
            self.areaRngLbl = ["all", "small", "medium", "large"]
        elif mode == "3D":
            self.iouThrs = np.linspace(0.05, 0.5, 10)
            self.areaRng = [[0, 1e5], [0, 10], [10, 35], [35, 1e5]]          # depth ranges (m)
            self.areaRngLbl = ["all", "near", "medium", "far"]
        else:
            raise Exception("mode %s not supported" % (mode))
        self.recThrs = np.linspace(0.0, 1.0, 101)
        self.maxDets = [1, 10, 100]
        self.useCats = 1
        self.iouType = "bbox"
        self.mode = mode
        self.proximity_thresh = 0.3
        self.imgIds, self.catIds = [], []


def iou_xywh(d, g):
    """pycocotools maskUtils.iou for boxes (iscrowd = 0): d (D,4), g (G,4) in [x,y,w,h] -> (D,G)."""
    d, g = np.asarray(d, np.float64).reshape(-1, 4), np.asarray(g, np.float64).reshape(-1, 4)
    x1 = np.maximum(d[:, None, 0], g[None, :, 0])
    y1 = np.maximum(d[:, None, 1], g[None, :, 1])
    x2 = np.minimum(d[:, None, 0] + d[:, None, 2], g[None, :, 0] + g[None, :, 2])
    y2 = np.minimum(d[:, None, 1] + d[:, None, 3], g[None, :, 1] + g[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    union = (d[:, 2] * d[:, 3])[:, None] + (g[:, 2] * g[:, 3])[None, :] - inter
    return np.where(union > 0, inter / np.where(union > 0, union, 1), 0.0)


def _iou3d_device(d, g):
    """(D,8,3), (G,8,3) -> (D,G) through cr_box3d_overlap on cuda:0 (the product path: no CPU fallback)."""
    import torch
    from ... import geometry
    dev = torch.device("cuda:0")
    _, iou = geometry.box3d_overlap(torch.as_tensor(np.asarray(d), dtype=torch.float32, device=dev),
                                    torch.as_tensor(np.asarray(g), dtype=torch.float32, device=dev))
    return iou.cpu().numpy().astype(np.float64)


class Omni3Deval:
    def __init__(self, gts, dts, mode="2D", eval_prox=False, iou3d_fn=None):
        if mode not in ("2D", "3D"):
            raise Exception("{} mode is not supported".format(mode))
        self.mode, self.eval_prox = mode, eval_prox
        self.params = Omni3DParams(mode)
        self.iou3d_fn = iou3d_fn or _iou3d_device
        self._gts, self._dts = collections.defaultdict(list), collections.defaultdict(list)
        for g in gts:
            self._gts[g["image_id"], g["category_id"]].append(dict(g))
        for i, d in enumerate(dts):
            d = dict(d)
            d.setdefault("id", i + 1)
            self._dts[d["image_id"], d["category_id"]].append(d)
        self.params.imgIds = sorted({k[0] for k in list(self._gts) + list(self._dts)})
        self.params.catIds = sorted({k[1] for k in list(self._gts) + list(self._dts)})
        self.evalImgs, self.eval, self.stats = {}, None, None

    # ------------------------------------------------------------------ per image / category
    def _ious(self, img, cat):
        gt, dt = self._gts[img, cat], self._dts[img, cat]
        order = np.argsort([-d["score"] for d in dt], kind="mergesort")[:self.params.maxDets[-1]]
        dt = [dt[i] for i in order]
        if not gt or not dt:
            return dt, np.zeros((len(dt), len(gt))), None
        key = "bbox" if self.mode == "2D" else "bbox3D"
        if self.mode == "2D":
            ious = iou_xywh([d[key] for d in dt], [g[key] for g in gt])
        else:
            ious = self.iou3d_fn([d[key] for d in dt], [g[key] for g in gt])
        prox = iou_xywh([d["bbox"] for d in dt], [g["bbox"] for g in gt]) > self.params.proximity_thresh if self.eval_prox else None
        return dt, ious, prox

    def _evaluate_img(self, img, cat, rng, max_det, dt_sorted, ious, prox):
        p = self.params
        gt = self._gts[img, cat]
        if not gt and not dt_sorted:
            return None
        f_rng = "area" if self.mode == "2D" else "depth"
        f_ign = "ignore2D" if self.mode == "2D" else "ignore3D"
        g_ign = np.array([int(bool(g.get(f_ign, 0)) or g[f_rng] < rng[0] or g[f_rng] > rng[1]) for g in gt], dtype=np.int64)
        gorder = np.argsort(g_ign, kind="mergesort")                       # ignored ground truth last
        g_ign = g_ign[gorder]
        dt = dt_sorted[:max_det]
        D, G, T = len(dt), len(gt), len(p.iouThrs)
        iou = ious[:D][:, gorder] if D and G else np.zeros((D, G))
        px = prox[:D][:, gorder] if (prox is not None and D and G) else None
        dtm, gtm, dt_ig = np.zeros((T, D)), np.zeros((T, G)), np.zeros((T, D), dtype=bool)
        for t, thr in enumerate(p.iouThrs):
            for di in range(D):
                best, m = min(thr, 1 - 1e-10), -1
                for gi in range(G):
                    if px is not None and not px[di, gi]:
                        continue
                    if gtm[t, gi] > 0:
                        continue
                    if m > -1 and g_ign[m] == 0 and g_ign[gi] == 1:
                        break                                               # a real match is never traded for an ignored one
                    if iou[di, gi] < best:
                        continue
                    best, m = iou[di, gi], gi
                if m >= 0:
                    dt_ig[t, di] = bool(g_ign[m])
                    dtm[t, di] = gt[gorder[m]]["id"]
                    gtm[t, m] = dt[di]["id"]
        out = np.array([d[f_rng] < rng[0] or d[f_rng] > rng[1] for d in dt], dtype=bool).reshape(1, D)
        dt_ig = dt_ig | ((dtm == 0) & out)                                  # unmatched detections outside the range
        if px is not None and D and G:
            dt_ig = dt_ig | (~px.any(1)).reshape(1, D)
        return {"dtMatches": dtm, "dtIgnore": dt_ig, "gtIgnore": g_ign, "dtScores": np.array([d["score"] for d in dt])}

    def evaluate(self):
        p = self.params
        self.evalImgs = {}
        for cat in p.catIds:
            for img in p.imgIds:
                dt_sorted, ious, prox = self._ious(img, cat)
                for a, rng in enumerate(p.areaRng):
                    self.evalImgs[cat, a, img] = self._evaluate_img(img, cat, rng, p.maxDets[-1], dt_sorted, ious, prox)
        return self

    # ------------------------------------------------------------------ dataset level
    def accumulate(self):
        p = self.params
        T, R, K, A, M = len(p.iouThrs), len(p.recThrs), len(p.catIds), len(p.areaRng), len(p.maxDets)
        precision, recall = -np.ones((T, R, K, A, M)), -np.ones((T, K, A, M))
        for k, cat in enumerate(p.catIds):
            for a in range(A):
                E = [e for e in (self.evalImgs[cat, a, img] for img in p.imgIds) if e is not None]
                if not E:
                    continue
                g_ign = np.concatenate([e["gtIgnore"] for e in E])
                npig = int((g_ign == 0).sum())
                if npig == 0:
                    continue
                for m, max_det in enumerate(p.maxDets):
                    scores = np.concatenate([e["dtScores"][:max_det] for e in E])
                    order = np.argsort(-scores, kind="mergesort")
                    dtm = np.concatenate([e["dtMatches"][:, :max_det] for e in E], axis=1)[:, order]
                    dti = np.concatenate([e["dtIgnore"][:, :max_det] for e in E], axis=1)[:, order]
                    tp = np.cumsum((dtm != 0) & ~dti, axis=1).astype(float)
                    fp = np.cumsum((dtm == 0) & ~dti, axis=1).astype(float)
                    for t in range(T):
                        nd = tp.shape[1]
                        rc = tp[t] / npig
                        pr = tp[t] / (fp[t] + tp[t] + np.spacing(1))
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        pr = np.maximum.accumulate(pr[::-1])[::-1] if nd else pr       # precision envelope
                        idx = np.searchsorted(rc, p.recThrs, side="left")
                        q = np.zeros(R)
                        ok = idx < nd
                        q[ok] = pr[idx[ok]]
                        precision[t, :, k, a, m] = q
        self.eval = {"precision": precision, "recall": recall, "counts": [T, R, K, A, M]}
        return self

    def _mean(self, ap, iou_thr=None, rng="all", max_det=100):
        p = self.params
        a = p.areaRngLbl.index(rng)
        m = p.maxDets.index(max_det)
        s = self.eval["precision"][:, :, :, a, m] if ap else self.eval["recall"][:, :, a, m]
        if iou_thr is not None:
            s = s[np.isclose(iou_thr, p.iouThrs)]
        s = s[s > -1]
        return float(s.mean()) if s.size else -1.0

    def summarize(self):
        """stats[0..12] in the reference's order: AP, AP@thr1..3, AP by range (3), AR@1/10/100, AR by range (3)."""
        if self.eval is None:
            raise Exception("Please run accumulate() first")
        p = self.params
        thr = [0.5, 0.75, 0.95] if self.mode == "2D" else [0.15, 0.25, 0.50]
        L = p.areaRngLbl
        self.stats = np.array([self._mean(1)] + [self._mean(1, iou_thr=t) for t in thr] +
                              [self._mean(1, rng=L[i]) for i in (1, 2, 3)] +
                              [self._mean(0, max_det=d) for d in p.maxDets] + [self._mean(0, rng=L[i]) for i in (1, 2, 3)])
        return self.stats
