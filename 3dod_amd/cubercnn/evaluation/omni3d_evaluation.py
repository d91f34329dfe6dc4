"""AP2D / AP3D evaluation -- the scoring core of cubercnn/evaluation/omni3d_evaluation.py of the reference
(Omni3DParams :1020-1088, Omni3Deval.computeIoU :1360-1432, evaluateImg :1434-1553, accumulate :1173-1315,
summarize :1555-1700; the protocol is pycocotools' COCOeval [third-party], restated).

Scope: in-memory ground truth / detection records -> the 13 summary numbers of each mode.  3D IoU goes
through the exact-IoU kernel (geometry.box3d_overlap = cr_box3d_overlap) instead of pytorch3d.  The dataset plumbing on
top of it (instances_to_coco_json, Omni3DEvaluator, Omni3DEvaluationHelper, inference_on_dataset) is at the end of the
file; the logging tables of utils_logperf are not built.

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
            if "area" not in d and "bbox" in d:
                d["area"] = d["bbox"][2] * d["bbox"][3]
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


# ---------------------------------------------------------------------------------------------------------------------
# dataset plumbing (reference: Omni3DEvaluator :644-935, Omni3DEvaluationHelper :168-520, inference_on_dataset :523-641,
# instances_to_coco_json :971-1014).  Result tables / histograms (utils_logperf) are logging only and not built.
# ---------------------------------------------------------------------------------------------------------------------
METRICS = {"2D": ["AP", "AP50", "AP75", "AP95", "APs", "APm", "APl"],
           "3D": ["AP", "AP15", "AP25", "AP50", "APn", "APm", "APf"]}


def instances_to_coco_json(instances, img_id):
    """:971-1014: CPU `Instances` of one image -> list of result records (bbox XYWH; `depth` = mean z of the 8 corners;
    unit placeholders when the model has no 3D head)."""
    n = len(instances)
    if n == 0:
        return []
    b = instances.pred_boxes.tensor.numpy().astype(np.float64).copy()
    b[:, 2:] -= b[:, :2]
    boxes, scores, classes = b.tolist(), instances.scores.tolist(), instances.pred_classes.tolist()
    if instances.has("pred_bbox3D"):
        bbox3D, center_cam = instances.pred_bbox3D.tolist(), instances.pred_center_cam.tolist()
        center_2D, dims, pose = instances.pred_center_2D.tolist(), instances.pred_dimensions.tolist(), instances.pred_pose.tolist()
    else:
        bbox3D, center_cam = np.ones([n, 8, 3]).tolist(), np.ones([n, 3]).tolist()
        center_2D, dims, pose = np.ones([n, 2]).tolist(), np.ones([n, 3]).tolist(), np.ones([n, 3, 3]).tolist()
    return [{"image_id": img_id, "category_id": classes[k], "bbox": boxes[k], "score": scores[k],
             "depth": float(np.array(bbox3D[k])[:, 2].mean()), "bbox3D": bbox3D[k], "center_cam": center_cam[k],
             "center_2D": center_2D[k], "dimensions": dims[k], "pose": pose[k]} for k in range(n)]


def _per_category(ev, names):
    """AP x100 per category from precision[T,R,K,A,M] at area 'all', maxDets[-1]"""
    out = {}
    pr = ev.eval["precision"]
    assert len(names) == pr.shape[2], (len(names), pr.shape)
    for k, name in enumerate(names):
        s = pr[:, :, k, 0, -1]
        s = s[s > -1]
        out["AP-" + name] = float(s.mean() * 100) if s.size else float("nan")
    return out


def _headline(ev, mode):
    return {m: float(ev.stats[i] * 100) if ev.stats[i] >= 0 else float("nan") for i, m in enumerate(METRICS[mode])}


class Omni3DEvaluator:
    """:644-935.  Ground truth = the dataset's json through `Omni3D([json], filter_settings)`; predictions arrive with
    the MODEL's contiguous class ids, are mapped back to Omni3D category ids and restricted to the categories this
    dataset annotates."""

    def __init__(self, dataset_name, tasks=None, distributed=True, output_dir=None, *, max_dets_per_image=None,
                 use_fast_impl=False, eval_prox=False, only_2d=False, filter_settings=None, iou3d_fn=None):
        from ...d2lite.data import MetadataCatalog
        from ..data.datasets import Omni3D
        self._output_dir, self._eval_prox, self._only_2d = output_dir, eval_prox, only_2d
        self._filter_settings, self._iou3d_fn = filter_settings, iou3d_fn
        self._max_dets_per_image = [1, 10, 100 if max_dets_per_image is None else max_dets_per_image]
        self._metadata = MetadataCatalog.get(dataset_name)
        self._omni_api = Omni3D([self._metadata.json_file], filter_settings)
        self._do_evaluation = "annotations" in self._omni_api.dataset
        self._predictions, self._results, self.evals = [], {}, {}
        if self._metadata.get("thing_classes") is None:
            # normally filled when the dataset is first loaded (load_omni3d_json, datasets.py:375-383); evaluating
            # stored predictions without building a loader must not depend on that having happened
            cats = sorted(self._omni_api.dataset["categories"], key=lambda c: c["id"])
            self._metadata.thing_classes = [c["name"] for c in cats]
            self._metadata.thing_dataset_id_to_contiguous_id = \
                MetadataCatalog.get('omni3d_model').thing_dataset_id_to_contiguous_id

    def reset(self):
        self._predictions = []

    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            pred = {"image_id": inp["image_id"], "K": inp["K"], "width": inp["width"], "height": inp["height"]}
            if "p2" in inp:
                pred["p2"] = inp["p2"]
            inst = out["instances"]
            pred["instances"] = inst if type(inst) == list else instances_to_coco_json(inst.to("cpu"), inp["image_id"])
            self._predictions.append(pred)

    def evaluate(self, img_ids=None):
        import copy
        import itertools
        import json
        import os
        from ...d2lite.data import MetadataCatalog
        self._results = {}
        if len(self._predictions) == 0:
            return {}
        results = copy.deepcopy(list(itertools.chain(*[p["instances"] for p in self._predictions])))
        global_names = MetadataCatalog.get('omni3d_model').thing_classes
        id_map = self._metadata.thing_dataset_id_to_contiguous_id
        contiguous = list(id_map.values())
        assert min(contiguous) == 0 and max(contiguous) == len(contiguous) - 1
        reverse = {v: k for k, v in id_map.items()}
        kept = []
        for r in results:
            c = r["category_id"]
            assert c < len(contiguous), f"A prediction has class={c}, but the dataset only has {len(contiguous)} classes"
            r["category_id"] = reverse[c]
            if global_names[c] in self._metadata.thing_classes:       # out-of-vocabulary for this dataset: dropped
                kept.append(r)
        if self._output_dir:
            os.makedirs(self._output_dir, exist_ok=True)
            with open(os.path.join(self._output_dir, "omni_instances_results.json"), "w") as f:
                json.dump(kept, f)
        if not self._do_evaluation or not kept:
            return copy.deepcopy(self._results)
        gts = self._omni_api.dataset["annotations"]
        for i, r in enumerate(kept):                                   # pycocotools loadRes for boxes
            r["area"], r["id"], r["iscrowd"] = r["bbox"][2] * r["bbox"][3], i + 1, 0
        cat_ids, all_imgs = sorted(self._omni_api.getCatIds()), sorted(self._omni_api.getImgIds())
        for mode in (["2D"] if self._only_2d else ["2D", "3D"]):
            ev = Omni3Deval(gts, kept, mode=mode, eval_prox=self._eval_prox, iou3d_fn=self._iou3d_fn)
            ev.params.catIds = cat_ids
            ev.params.imgIds = list(img_ids) if img_ids is not None else all_imgs
            ev.params.maxDets = list(self._max_dets_per_image)
            ev.evaluate().accumulate().summarize()
            res = _headline(ev, mode)
            names = self._metadata.get("thing_classes")
            if names is not None and len(names) > 1:
                res.update(_per_category(ev, names))
            self._results["bbox_" + mode] = res
            self.evals[mode] = ev
        return copy.deepcopy(self._results)


class Omni3DEvaluationHelper:
    """:168-520: one evaluator per dataset split, plus the pooled ("<Concat>") numbers and the Omni3D / indoor /
    outdoor averages, computed from the per-image evaluations already made (no IoU is recomputed)."""

    def __init__(self, dataset_names, filter_settings, output_folder, iter_label='-', only_2d=False, iou3d_fn=None):
        import os
        from collections import OrderedDict
        from ...d2lite.data import MetadataCatalog
        from ..data.datasets import simple_register
        self.dataset_names, self.filter_settings, self.output_folder = dataset_names, filter_settings, output_folder
        self.iter_label, self.only_2d = iter_label, only_2d
        self.evaluators, self.results = OrderedDict(), OrderedDict()
        self.results_analysis, self.results_omni3d = OrderedDict(), OrderedDict()
        self.overall_imgIds, self.overall_catIds = set(), set()
        self.output_folders = {n: os.path.join(output_folder, n) for n in dataset_names}
        for n in dataset_names:
            if MetadataCatalog.get(n).get('json_file') is None:
                simple_register(n, filter_settings, filter_empty=False)
            ev = Omni3DEvaluator(n, output_dir=self.output_folders[n], filter_settings=filter_settings, only_2d=only_2d,
                                 eval_prox=('Objectron' in n or 'SUNRGBD' in n), distributed=False, iou3d_fn=iou3d_fn)
            ev.reset()
            self.evaluators[n] = ev
            self.overall_imgIds.update(ev._omni_api.getImgIds())
            self.overall_catIds.update(ev._omni_api.getCatIds())

    def add_predictions(self, dataset_name, predictions):
        self.evaluators[dataset_name]._predictions += predictions

    def save_predictions(self, dataset_name):
        import os
        import torch
        os.makedirs(self.output_folders[dataset_name], exist_ok=True)
        torch.save(self.evaluators[dataset_name]._predictions,
                   os.path.join(self.output_folders[dataset_name], "instances_predictions.pth"))

    def evaluate(self, dataset_name):
        from ..data.builtin import get_omni3d_categories
        if dataset_name not in self.results:
            self.results[dataset_name] = self.evaluators[dataset_name].evaluate()
        res = self.results[dataset_name]
        if "bbox_2D" not in res:
            return res
        names = self.filter_settings['category_names']
        cats = {c for c in names if 'AP-' + c in res['bbox_2D']}
        mean = lambda mode, cs: float(np.mean([res['bbox_' + mode]['AP-' + c] for c in cs])) if cs else float("nan")
        g2 = mean("2D", cats)
        g3 = float("nan") if self.only_2d else mean("3D", cats)
        o2 = o3 = float("nan")
        try:
            own = get_omni3d_categories(dataset_name)
        except ValueError:
            own = None
        if own is not None and len(own - cats) == 0:
            o2 = mean("2D", own)
            o3 = float("nan") if self.only_2d else mean("3D", own)
        self.results_omni3d[dataset_name] = {"iters": self.iter_label, "AP2D": o2, "AP3D": o3}
        r3 = res.get('bbox_3D', {})
        nan = float("nan")
        self.results_analysis[dataset_name] = {
            "iters": self.iter_label, "AP2D": g2, "AP3D": g3, "AP3D@15": r3.get('AP15', nan), "AP3D@25": r3.get('AP25', nan),
            "AP3D@50": r3.get('AP50', nan), "AP3D-N": r3.get('APn', nan), "AP3D-M": r3.get('APm', nan), "AP3D-F": r3.get('APf', nan)}
        return res

    def _pooled(self, mode):
        """accumulate over the union of all datasets' per-image evaluations"""
        ev = Omni3Deval([], [], mode=mode)
        ev.params.catIds, ev.params.imgIds = list(self.overall_catIds), list(self.overall_imgIds)
        ev.evalImgs = {}
        for e in self.evaluators.values():
            if mode in e.evals:
                ev.evalImgs.update({k: v for k, v in e.evals[mode].evalImgs.items() if v is not None})
        for c in ev.params.catIds:
            for a in range(len(ev.params.areaRng)):
                for i in ev.params.imgIds:
                    ev.evalImgs.setdefault((c, a, i), None)
        ev.accumulate().summarize()
        return ev

    def summarize_all(self):
        from ...d2lite.data import MetadataCatalog
        from ..data.builtin import get_omni3d_categories
        for n in self.dataset_names:
            if n not in self.results:
                self.evaluate(n)
        meta = MetadataCatalog.get('omni3d_model')
        ordered = [meta.thing_classes[meta.thing_dataset_id_to_contiguous_id[c]] for c in self.overall_catIds]
        cats = set(ordered)
        res = {}
        for mode in (["2D"] if self.only_2d else ["2D", "3D"]):
            ev = self._pooled(mode)
            res[mode] = {**_headline(ev, mode), **_per_category(ev, ordered)}
        nan = float("nan")
        mean = lambda mode, cs: float(np.mean([res[mode]['AP-' + c] for c in cs])) if mode in res else nan
        r3 = res.get("3D", {})
        self.results_analysis["<Concat>"] = {
            "iters": self.iter_label, "AP2D": mean("2D", cats), "AP3D": mean("3D", cats), "AP3D@15": r3.get('AP15', nan),
            "AP3D@25": r3.get('AP25', nan), "AP3D@50": r3.get('AP50', nan), "AP3D-N": r3.get('APn', nan),
            "AP3D-M": r3.get('APm', nan), "AP3D-F": r3.get('APf', nan)}
        for label, key in (("Omni3D_Out", "omni3d_out"), ("Omni3D_In", "omni3d_in"), ("Omni3D", "omni3d")):
            group = get_omni3d_categories(key)
            full = len(group - cats) == 0
            self.results_omni3d[label] = {"iters": self.iter_label, "AP2D": mean("2D", group) if full else nan,
                                          "AP3D": mean("3D", group) if full else nan}
        self.results_concat = res
        return self.results_analysis, self.results_omni3d


def inference_on_dataset(model, data_loader):
    """:523-641: runs `model` (eval mode, no grad) over this rank's loader and returns the prediction records; with
    torch.distributed initialised the records of all ranks are gathered on rank 0 (other ranks get [])."""
    import itertools
    import torch
    import torch.distributed as dist
    import logging
    import time
    was_training = getattr(model, "training", False)
    if hasattr(model, "eval"):
        model.eval()
    if hasattr(model, "enable_graphs_eval") and next(iter(model.parameters())).is_cuda:
        model.enable_graphs_eval(max_shapes=16)          # one forward graph per image resolution of the dataset
    out = []
    t_data = t_compute = t_eval = 0.0
    n_iter = 0
    try:
        with torch.no_grad():
            t0 = time.perf_counter()
            for inputs in data_loader:
                t1 = time.perf_counter()
                outputs = model(inputs)
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                t2 = time.perf_counter()
                for inp, o in zip(inputs, outputs):
                    out.append({"image_id": inp["image_id"], "K": inp["K"], "width": inp["width"], "height": inp["height"],
                                "instances": instances_to_coco_json(o["instances"].to("cpu"), inp["image_id"])})
                t3 = time.perf_counter()
                t_data, t_compute, t_eval, n_iter = t_data + (t1 - t0), t_compute + (t2 - t1), t_eval + (t3 - t2), n_iter + 1
                t0 = t3
    finally:
        if was_training:
            model.train()
    if n_iter:
        # the reference's accounting (:549-632): data / compute / packing seconds per iteration on this rank
        logging.getLogger(__name__).info(
            "Inference done {} iterations. Dataloading: {:.4f} s/iter. Inference: {:.4f} s/iter. Eval: {:.4f} s/iter.".format(
                n_iter, t_data / n_iter, t_compute / n_iter, t_eval / n_iter))
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        gathered = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
        dist.gather_object(out, gathered, dst=0)
        if dist.get_rank() != 0:
            return []
        out = list(itertools.chain(*gathered))
    return out
