"""RCNN3D meta-architecture -- cubercnn/modeling/meta_arch/rcnn3d.py:34-124,894-918 of the reference on a
detectron2 GeneralizedRCNN stand-in [third-party: preprocess_image, _postprocess].
model(list[dict]) -> dict of scalar losses (training) or list[{"instances": Instances}] (eval)."""
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from ....d2lite import META_ARCH_REGISTRY, BACKBONE_REGISTRY, ImageList, Instances, ShapeSpec, Boxes, get_event_storage
from .... import hipops as ops
from ..proposal_generator import build_proposal_generator
from ..roi_heads import build_roi_heads


def detector_postprocess(results: Instances, output_height: int, output_width: int):
    """detectron2 detector_postprocess [third-party]: rescale 2D boxes to the requested output size."""
    scale_x, scale_y = output_width / results.image_size[1], output_height / results.image_size[0]
    out = Instances((output_height, output_width), **results.get_fields())
    if out.has("pred_boxes"):
        b = out.pred_boxes.clone()
        b.scale(scale_x, scale_y)
        b.tensor = torch.stack((b.tensor[:, 0].clamp(0, output_width), b.tensor[:, 1].clamp(0, output_height),
                                b.tensor[:, 2].clamp(0, output_width), b.tensor[:, 3].clamp(0, output_height)), -1)
        out.pred_boxes = b
        out = out[b.nonempty()]
    return out


@META_ARCH_REGISTRY.register()
class RCNN3D(nn.Module):
    def __init__(self, cfg, priors=None):
        super().__init__()
        self.backbone = build_backbone(cfg, priors=priors)
        self.proposal_generator = build_proposal_generator(cfg, self.backbone.output_shape())
        self.roi_heads = build_roi_heads(cfg, self.backbone.output_shape(), priors=priors)
        self.input_format = cfg.INPUT.FORMAT
        self.vis_period = cfg.VIS_PERIOD
        self.pixel_mean_list = [float(v) for v in cfg.MODEL.PIXEL_MEAN]
        self.pixel_std_list = [float(v) for v in cfg.MODEL.PIXEL_STD]
        # bf16 compute copies of the weights are cached per "weight epoch" (hipops): loading a checkpoint moves it
        self.register_load_state_dict_post_hook(lambda module, incompatible: ops.bump_weight_epoch())
        self.register_buffer("pixel_mean", torch.tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1), False)
        self._graphed = None
        self._graphed_cache, self._graphed_max, self._graphed_split = None, 0, False
        self._graphed_eval = None
        self._graphed_eval_cache, self._graphed_eval_max = None, 0
        # static-shape training path (modeling/dense_train.py): same rules, no host<->device syncs
        self.dense_train = True

    def enable_graphs(self, sample_batched_inputs=None, split_backward=False, max_shapes=0):
        """capture the static dense region (trunk + FPN + RPN head, forward and backward) as HIP graphs.  Call after the
        optimizer (FlatSGD) has been built.
        sample_batched_inputs: capture now for this image-batch shape.
        max_shapes > 0: additionally keep one captured region per image-batch shape met in training, captured on first sight
        and kept for the `max_shapes` most recently used shapes (what `do_train` turns on: with INPUT.MIN_SIZE_TRAIN a run
        meets a handful of padded resolutions).  A batch whose images differ in size, or any shape beyond the cache when
        max_shapes == 0, runs eagerly.
        split_backward: two backward graphs, so that a data-parallel step all-reduces the first segment's gradients under
        the second (graphed.GraphedDense)."""
        from collections import OrderedDict
        from ..graphed import GraphedDense
        self._graphed_max, self._graphed_split = int(max_shapes), bool(split_backward)
        if self._graphed_cache is None:
            self._graphed_cache = OrderedDict()
        if sample_batched_inputs is not None:
            il, batch = self._stack_images(sample_batched_inputs)
            self._graphed = GraphedDense(self, batch, split_backward=split_backward)
            if self._graphed_max > 0:
                self._graphed_cache[(self._graphed.shape, ops.precision())] = self._graphed
        return self._graphed

    def disable_graphs(self):
        """back to eager launches of the dense region (captured graphs are dropped)"""
        self._graphed, self._graphed_cache, self._graphed_max = None, None, 0

    def _train_graph_for(self, batch):
        """the captured dense region for this uint8 image batch: the one used last, one from the per-shape cache, or a new
        capture if the cache is on (least recently used shape evicted)"""
        g = self._graphed
        if g is not None and g.matches(batch):
            return g
        cache = self._graphed_cache
        if not self._graphed_max or cache is None:
            return None
        key = (tuple(batch.shape), ops.precision())
        g = cache.get(key)
        if g is not None and not g.matches(batch):            # e.g. the optimizer's weight bank was rebuilt: capture again
            del cache[key]
            g = None
        if g is None:
            from ..graphed import GraphedDense
            while len(cache) >= self._graphed_max:
                cache.popitem(last=False)
            g = cache[key] = GraphedDense(self, batch, split_backward=self._graphed_split)
        else:
            cache.move_to_end(key)
        self._graphed = g
        return g

    def enable_graphs_eval(self, sample_batched_inputs=None, max_shapes=0):
        """eval-mode counterpart of enable_graphs: forward-only graph of preprocess + trunk + FPN + RPN head for the
        image-batch shape of `sample_batched_inputs`, and / or (max_shapes > 0) one graph per batch shape met during
        inference, captured on first sight and kept for the `max_shapes` most recently used shapes -- an evaluation set has
        a handful of resolutions, and at one image per step the eager dense region is bound by the host's launch rate."""
        from collections import OrderedDict
        from ..graphed import GraphedDenseEval
        self._graphed_eval_max = int(max_shapes)
        if self._graphed_eval_cache is None:
            self._graphed_eval_cache = OrderedDict()
        if sample_batched_inputs is not None:
            il, batch = self._stack_images(sample_batched_inputs)
            self._graphed_eval = GraphedDenseEval(self, batch)
            self._graphed_eval_cache[(self._graphed_eval.shape, ops.precision())] = self._graphed_eval
        return self._graphed_eval

    def _eval_graph_for(self, batch):
        """the captured dense region for this uint8 image batch, capturing it if the per-shape cache has room"""
        cache = self._graphed_eval_cache
        if cache is None:
            return None
        key = (tuple(batch.shape), ops.precision())
        ge = cache.get(key)
        if ge is not None:
            cache.move_to_end(key)
            return ge if ge.matches(batch) else None
        if self._graphed_eval_max <= 0:
            return None
        from ..graphed import GraphedDenseEval
        while len(cache) >= self._graphed_eval_max:
            cache.popitem(last=False)                       # least recently used shape
        ge = cache[key] = GraphedDenseEval(self, batch)
        return ge

    def forward_static(self, images_u8, image_sizes, gt, meta):
        """training forward from device-resident, fixed-shape inputs only (no host data, no syncs): the body of the
        whole-step HIP graph (solver.GraphedTrainStep).  images_u8 (B,3,H,W) uint8; gt: dense_train.GTBatch; meta (B,5)."""
        from ..dense_train import forward_train
        x = ops.preprocess(images_u8, self.pixel_mean_list, self.pixel_std_list)
        features = self.backbone(x)
        return forward_train(self, image_sizes, features, None, gt, meta)          # (runs the RPN head in its raw form)

    def _stack_images(self, batched_inputs):
        images = [x["image"].to(self.device) for x in batched_inputs]
        il = ImageList.from_tensors(images, self.backbone.size_divisibility,
                                    padding_constraints=self.backbone.padding_constraints)
        batch = il.tensor
        if batch.dtype != torch.uint8:
            batch = batch.clamp(0, 255).to(torch.uint8)
        return il, batch.contiguous()

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess_image(self, batched_inputs):
        """(x - mean)/std, pad to the backbone's size divisibility, stack.  Returns (ImageList of the uint8 batch
        sizes, NHWC bf16 tensor with 8 channels)."""
        il, batch = self._stack_images(batched_inputs)
        x = ops.preprocess(batch, self.pixel_mean_list, self.pixel_std_list)
        H, W = batch.shape[-2:]
        if any(tuple(s) != (H, W) for s in il.image_sizes):
            # detectron2 pads AFTER normalisation (zeros in normalised space)
            mask = torch.zeros((len(il.image_sizes), H, W, 1), dtype=x.dtype, device=x.device)
            for i, (h, w) in enumerate(il.image_sizes):
                mask[i, :h, :w] = 1
            x = x * mask
        return il, x

    def forward(self, batched_inputs: List[Dict[str, torch.Tensor]]):
        if not self.training:
            return self.inference(batched_inputs)
        head_outputs = None
        g = None
        if self._graphed is not None or self._graphed_max:
            images, batch = self._stack_images(batched_inputs)
            if all(tuple(sz) == tuple(batch.shape[-2:]) for sz in images.image_sizes):
                g = self._train_graph_for(batch)
            if g is not None:
                from ..dense_train import RawRPNOutputs
                features, ys = g(batch)
                head_outputs = RawRPNOutputs(ys)
        if g is None:
            images, x = self.preprocess_image(batched_inputs)
            features = self.backbone(x)
        im_scales_ratio = [info['height'] / im_size[0] for (info, im_size) in zip(batched_inputs, images.image_sizes)]
        Ks = [torch.FloatTensor(info['K']) for info in batched_inputs]
        if "instances" in batched_inputs[0]:
            gt_instances = [b["instances"].to(self.device) for b in batched_inputs]
        else:
            gt_instances = None
        if self.dense_train and gt_instances is not None:
            return self._forward_dense(images, features, head_outputs, gt_instances, Ks, im_scales_ratio, batched_inputs)
        if head_outputs is not None:                       # instance-list path (tests): the lists of RPN.forward
            pg = self.proposal_generator
            A, D = pg.rpn_head.num_anchors, pg.rpn_head.box_dim
            ys = pg.rpn_head.level_views(head_outputs.ys, [features[f] for f in pg.in_features])
            head_outputs = ([y[..., :A].reshape(y.shape[0], -1) for y in ys],
                            [y[..., A:A + A * D].reshape(y.shape[0], -1, D) for y in ys])
        proposals, proposal_losses = self.proposal_generator(images, features, gt_instances, head_outputs=head_outputs)
        instances, detector_losses = self._run_roi_heads(images, features, proposals, Ks, im_scales_ratio, gt_instances,
                                                         batched_inputs)
        losses = {}
        losses.update(detector_losses)
        losses.update(proposal_losses)
        return losses

    def _run_roi_heads(self, images, features, proposals, Ks, im_scales_ratio, targets, batched_inputs):
        return self.roi_heads(images, features, proposals, Ks, im_scales_ratio, targets)

    def _forward_dense(self, images, features, head_outputs, gt_instances, Ks, im_scales_ratio, batched_inputs):
        """static-shape, sync-free training forward (modeling/dense_train.py)"""
        from ..dense_train import forward_train, GTBatch, camera_meta
        dev = self.device
        im_dims = [tuple(s) for s in images.image_sizes]
        return forward_train(self, im_dims, features, head_outputs, GTBatch(gt_instances, dev),
                             camera_meta(self.roi_heads, Ks, im_scales_ratio, im_dims, dev))

    def inference(self, batched_inputs, detected_instances=None, do_postprocess: bool = True):
        assert not self.training
        head_outputs = None
        ge = None
        if self._graphed_eval_cache is not None:
            images, batch = self._stack_images(batched_inputs)
            if all(tuple(sz) == tuple(batch.shape[-2:]) for sz in images.image_sizes):
                ge = self._eval_graph_for(batch)
            if ge is not None:
                features, logits, deltas = ge(batch)
                head_outputs = (logits, deltas)
        if ge is None:
            images, x = self.preprocess_image(batched_inputs)
            features = self.backbone(x)
        im_scales_ratio = [info['height'] / im_size[0] for (info, im_size) in zip(batched_inputs, images.image_sizes)]
        Ks = [torch.FloatTensor(info['K']) for info in batched_inputs]
        if type(batched_inputs == list) and np.any(['oracle2D' in b for b in batched_inputs]):
            oracles = [b['oracle2D'] for b in batched_inputs]
            results, _ = self._run_roi_heads(images, features, oracles, Ks, im_scales_ratio, None, batched_inputs)
        else:
            proposals, _ = self.proposal_generator(images, features, None, head_outputs=head_outputs, padded=True)
            results, _ = self._run_roi_heads(images, features, proposals, Ks, im_scales_ratio, None, batched_inputs)
        if do_postprocess:
            return RCNN3D._postprocess(results, batched_inputs, images.image_sizes)
        return results

    @staticmethod
    def _postprocess(instances, batched_inputs, image_sizes):
        """detectron2 GeneralizedRCNN._postprocess [third-party]: rescale the 2D boxes to the requested output size, clip,
        drop empty boxes.  Done for the whole batch with one host sync; the per-image path (detector_postprocess) is
        taken only when some box actually becomes empty."""
        sizes = [(inp.get("height", isz[0]), inp.get("width", isz[1])) for inp, isz in zip(batched_inputs, image_sizes)]
        if len(instances) > 1 and all(r.has("pred_boxes") for r in instances):
            counts = [len(r) for r in instances]
            dev = instances[0].pred_boxes.tensor.device
            per_img = torch.tensor([[w / r.image_size[1], h / r.image_size[0], float(w), float(h)]
                                    for r, (h, w) in zip(instances, sizes)], dtype=torch.float32, device=dev)
            row = per_img.repeat_interleave(torch.tensor(counts, device=dev), dim=0)               # (n,4) sx, sy, W, H
            b = torch.cat([r.pred_boxes.tensor for r in instances])
            scale = row[:, [0, 1, 0, 1]]
            limit = row[:, [2, 3, 2, 3]]
            b = torch.minimum((b * scale).clamp(min=0), limit)
            nonempty = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
            if bool(nonempty.all()):                                                              # the one host sync
                out = []
                for r, (h, w), bb in zip(instances, sizes, b.split(counts)):
                    f = dict(r.get_fields())
                    f["pred_boxes"] = Boxes(bb)                    # the same rows, rescaled: lengths unchanged
                    out.append({"instances": Instances._from_fields((h, w), f)})
                return out
        return [{"instances": detector_postprocess(r, h, w)} for r, (h, w) in zip(instances, sizes)]


@META_ARCH_REGISTRY.register()
class RCNN3D_combined_features(RCNN3D):
    """rcnn3d.py:266-456: Cube R-CNN trained without 3D labels -- the RoI heads (`ROIHeads3DScore`) additionally get the
    per-image metric depth map and ground mask (padded to one size, `ImageList` keeps the true sizes; an image without a
    ground mask carries the (1,1) dummy of :381-384).  The depth-feature concatenation of `cat_depth_features`
    (MODEL.DEPTH_ON, Depth-Anything backbone) is off in every shipped config and not built."""

    def __init__(self, cfg, priors=None):
        super().__init__(cfg, priors=priors)
        if cfg.MODEL.get("DEPTH_ON", False):
            raise NotImplementedError("MODEL.DEPTH_ON needs the Depth-Anything feature extractor, which is not built")
        self.depth_model = None
        self.only_2d = cfg.MODEL.ROI_CUBE_HEAD.LOSS_W_3D == 0.0
        # dense_train = True: RPN, sampling and the box head run on the fused static-shape path, the weak cube losses on the
        # compacted foreground RoIs; False: the instance-list path written like the reference (both are tested)

    def _maps(self, batched_inputs, key):
        return ImageList.from_tensors([b[key].to(self.device) for b in batched_inputs])

    def _scene_maps(self, batched_inputs):
        if self.only_2d:
            return None, None
        depth_maps = self._maps(batched_inputs, "depth_map")
        # the (1,1) dummy of an image without ground map lives on the device: a fresh host tensor would cost a blocking copy,
        # i.e. a wait for the previous step's kernels, in every step
        dummy = self.__dict__.get("_ground_dummy")
        if dummy is None or dummy.device != self.device:
            dummy = self.__dict__["_ground_dummy"] = torch.ones((1, 1), dtype=torch.int64, device=self.device)
        filled = [dict(b, ground_map=dummy) if b.get("ground_map") is None else b for b in batched_inputs]
        return self._maps(filled, "ground_map"), depth_maps

    def _images_raw(self, batched_inputs):
        """rcnn3d.py:369: the un-normalised RGB images, padded to one size (only needed by a segmentor)"""
        if not (self.training and getattr(self.roi_heads, "needs_masks", False)):
            return None
        return ImageList.from_tensors([b["image"].to(self.device)[[2, 1, 0]] for b in batched_inputs])

    def _run_roi_heads(self, images, features, proposals, Ks, im_scales_ratio, targets, batched_inputs):
        ground_maps, depth_maps = self._scene_maps(batched_inputs) if self.training else (None, None)
        return self.roi_heads(images, self._images_raw(batched_inputs), ground_maps, depth_maps, features, proposals, Ks,
                              im_scales_ratio, targets)

    def _forward_dense(self, images, features, head_outputs, gt_instances, Ks, im_scales_ratio, batched_inputs):
        from ..dense_train import forward_train_weak, GTBatch
        ground_maps, depth_maps = self._scene_maps(batched_inputs)
        im_dims = [tuple(s) for s in images.image_sizes]
        masks, keys = self.roi_heads.object_masks(self._images_raw(batched_inputs), gt_instances)
        return forward_train_weak(self, im_dims, features, head_outputs, GTBatch(gt_instances, self.device), Ks,
                                  im_scales_ratio, ground_maps, depth_maps, masks, keys)


@META_ARCH_REGISTRY.register()
class BoxNet(RCNN3D):
    """the proposal-and-scoring meta-architecture, rcnn3d.py:594-759 of the reference: 2D detections (or GT boxes) ->
    ROIHeads_Boxer.  batched_inputs additionally carry "depth_map" (H,W) and optionally "ground_map" (H,W) and
    "masks" (N,H,W) per image (the reference reads depth/ground .npz files and runs SAM-HQ for the masks)."""

    def forward(self, batched_inputs, experiment_type=None, proposal_function='propose'):
        """rcnn3d.py:678-713.  Training mode generates pseudo ground truth on the GT boxes (experiment_type['pseudo_gt'] =
        'learn' | 'pseudo'), eval mode is the AP path or, with experiment_type['output_recall_scores'], the MABO tuple."""
        experiment_type = experiment_type or {'use_pred_boxes': True}
        if self.training:
            experiment_type = dict(experiment_type, use_pred_boxes=False)
        return self.inference(batched_inputs, experiment_type=experiment_type, proposal_function=proposal_function,
                              do_postprocess=not (self.training or experiment_type.get('output_recall_scores', False)))

    def inference(self, batched_inputs, experiment_type=None, do_postprocess=True, generator=None, proposal_function='propose'):
        use_pred = (experiment_type or {}).get('use_pred_boxes', True)
        if experiment_type and (self.training or experiment_type.get('output_recall_scores', False)):
            do_postprocess = False                    # these branches return cubes / score tables, not detections to rescale
        if use_pred:
            images, x = self.preprocess_image(batched_inputs)
        else:
            # GT boxes: only the image sizes are used -- no stacking, no normalisation
            from types import SimpleNamespace
            images = SimpleNamespace(image_sizes=[tuple(b["image"].shape[-2:]) for b in batched_inputs])
        im_scales_ratio = [info['height'] / im_size[0] for (info, im_size) in zip(batched_inputs, images.image_sizes)]
        Ks = [torch.FloatTensor(info['K']) for info in batched_inputs]
        depth = torch.stack([b["depth_map"] for b in batched_inputs]).to(self.device).float()
        ground = torch.stack([b["ground_map"] for b in batched_inputs]).to(self.device) \
            if all(b.get("ground_map") is not None for b in batched_inputs) else None
        masks = [b.get("masks") for b in batched_inputs] if any("masks" in b for b in batched_inputs) else None
        if use_pred:
            features = self.backbone(x)
            proposals, _ = self.proposal_generator(images, features, None, padded=True)
        else:
            features = None
            proposals = [b["instances"] if b["instances"].gt_boxes.device == self.device else b["instances"].to(self.device)
                         for b in batched_inputs]
        results, _ = self.roi_heads(images, features, proposals, depth, ground, Ks, im_scales_ratio, masks=masks,
                                    use_pred_boxes=use_pred, generator=generator, proposal_function=proposal_function,
                                    experiment_type=experiment_type)
        if do_postprocess:
            return RCNN3D._postprocess(results, batched_inputs, images.image_sizes)
        return results


def build_model(cfg, priors=None):
    """rcnn3d.py:894-903."""
    meta_arch = cfg.MODEL.META_ARCHITECTURE
    model = META_ARCH_REGISTRY.get(meta_arch)(cfg, priors=priors)
    model.to(torch.device(cfg.MODEL.DEVICE))
    # .to() keeps channels_last storage of the conv weights
    return model


def build_backbone(cfg, input_shape=None, priors=None):
    """rcnn3d.py:905-918."""
    if input_shape is None:
        input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    backbone_name = cfg.MODEL.BACKBONE.NAME
    return BACKBONE_REGISTRY.get(backbone_name)(cfg, input_shape, priors)
