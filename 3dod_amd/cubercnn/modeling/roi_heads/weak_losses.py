"""Losses of the weakly supervised 3D head (2D boxes + depth / ground maps instead of 3D labels), batched for the
device.  Reference: `ROIHeads3DScore` in cubercnn/modeling/roi_heads/roi_heads.py -- pose_loss :1055-1074,
normal_vector_from_maps :1076-1149, z_loss :1151-1194, pseudo_gt_z_box_loss :1196-1232, dim_loss :1234-1254,
pseudo_gt_z_point_loss :1256-1279, normal_to_rotation :1306-1317.  The reference loops over RoIs / images in Python;
here every loss is one batched expression over all RoIs, the depth median of a box is a radix-select kernel
(`cr_box_median`), the ground plane a RANSAC kernel (`cr_ransac_plane`).  Quirks of the reference that change values
are kept and marked "as in the reference".

Third-party pieces restated from their public definitions [parity unpinned]: torchvision.ops.generalized_box_iou_loss,
pytorch3d.transforms.so3_relative_angle / so3_rotation_angle.
"""
import torch

from ...util import math_util as util


# ---------------------------------------------------------------------------------------------- third-party restated
def generalized_box_iou_loss(boxes1, boxes2, reduction="none", eps=1e-7):
    """torchvision.ops.generalized_box_iou_loss: 1 - GIoU of paired XYXY boxes."""
    boxes1, boxes2 = boxes1.float(), boxes2.float()
    x1, y1, x2, y2 = boxes1.unbind(dim=-1)
    x1g, y1g, x2g, y2g = boxes2.unbind(dim=-1)
    xk1, yk1 = torch.max(x1, x1g), torch.max(y1, y1g)
    xk2, yk2 = torch.min(x2, x2g), torch.min(y2, y2g)
    inter = torch.where((yk2 > yk1) & (xk2 > xk1), (xk2 - xk1) * (yk2 - yk1), torch.zeros_like(x1))
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
    iou = inter / (union + eps)
    xc1, yc1 = torch.min(x1, x1g), torch.min(y1, y1g)
    xc2, yc2 = torch.max(x2, x2g), torch.max(y2, y2g)
    area_c = (xc2 - xc1) * (yc2 - yc1)
    loss = 1 - (iou - (area_c - union) / (area_c + eps))
    if reduction == "mean":
        return loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
    if reduction == "sum":
        return loss.sum()
    return loss


def so3_rotation_angle(R, eps=1e-4, cos_angle=False, cos_bound=1e-4):
    """pytorch3d: angle (or its cosine) of rotation matrices from the trace; raises on an invalid trace."""
    if R.dim() != 3 or R.shape[1:] != (3, 3):
        raise ValueError("Input has to be a batch of 3x3 Tensors.")
    tr = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    if ((tr < -1.0 - eps) + (tr > 3.0 + eps)).any():
        raise ValueError("A matrix has trace outside valid range [-1-eps,3+eps].")
    c = (tr - 1.0) * 0.5
    if cos_angle:
        return c
    b = 1.0 - cos_bound
    return torch.acos(c.clamp(-b, b)) if cos_bound > 0.0 else torch.acos(c)


def so3_relative_angle(R1, R2, cos_angle=False, cos_bound=1e-4, eps=1e-4):
    return so3_rotation_angle(torch.bmm(R1, R2.permute(0, 2, 1)), cos_angle=cos_angle, cos_bound=cos_bound, eps=eps)


# ---------------------------------------------------------------------------------------------- projection
def _int_clamp_bounds(c):
    """Cubes.get_bube_corners (spaces.py:240-243): [int(-c/2+1), int(2c-1)]"""
    return int(-c / 2 + 1), int(c - 1 + c)


def clamp_bounds_of(clamp_dims, device):
    """(n,4) [x_lo, x_hi, y_lo, y_hi] for a list of n (c0, c1): x is clamped with c0, y with c1 (the reference passes
    (H, W) of the image here: roi_heads.py:1419-1423,1551)."""
    return torch.tensor([_int_clamp_bounds(c[0]) + _int_clamp_bounds(c[1]) for c in clamp_dims], dtype=torch.float32,
                        device=device)


def project_cubes_to_corners(cubes, K, bounds):
    """cubes (n,P,15), K (n,3,3), bounds (n,4) from clamp_bounds_of (or the list of (c0, c1) itself)
    -> clamped projected corners (n,P,8,2)."""
    n, P = cubes.shape[:2]
    if not isinstance(bounds, torch.Tensor):
        bounds = clamp_bounds_of(bounds, cubes.device)
    verts = util.get_cuboid_verts_faces(cubes[..., :6].reshape(-1, 6), cubes[..., 6:].reshape(-1, 3, 3))[0]
    Kr = K[:, None].expand(n, P, 3, 3).reshape(-1, 3, 3)
    pc = torch.matmul(Kr, verts.transpose(2, 1))
    pc = (pc[:, :2, :] / pc[:, 2, :].unsqueeze(-2)).transpose(2, 1).reshape(n, P, 8, 2)
    b = bounds.to(pc.dtype).view(n, 1, 1, 4)
    x = torch.clamp(pc[..., 0], b[..., 0], b[..., 1])
    y = torch.clamp(pc[..., 1], b[..., 2], b[..., 3])
    return torch.stack((x, y), dim=-1)


def corners_to_boxes(corners):
    """conversions.py:25-48: XYXY hull of the 8 projected corners, (n,P,8,2) -> (n,P,4)"""
    return torch.stack((corners[..., 0].min(-1).values, corners[..., 1].min(-1).values,
                        corners[..., 0].max(-1).values, corners[..., 1].max(-1).values), dim=-1)


# ---------------------------------------------------------------------------------------------- losses
def pose_alignment_loss(cube_pose, num_boxes_per_image):
    """:1055-1074: per image the mean over all pairs (i > j) of 1 - |cos of the relative rotation angle|, summed over
    images and divided by (number of single-box images + 1); None when every image has exactly one box.
    tr(Ri Rj^T) is the Gram matrix of the flattened rotations."""
    P = cube_pose.reshape(-1, 9)
    total = torch.zeros(1, device=cube_pose.device, dtype=cube_pose.dtype)
    fail, start = 0, 0
    for m in num_boxes_per_image:
        blk = P[start:start + m]
        start += m
        if m == 1:
            fail += 1
            continue
        ij = torch.tril_indices(m, m, -1, device=P.device)
        cos = ((blk[ij[0]] * blk[ij[1]]).sum(1) - 1.0) * 0.5
        total = total + torch.mean(1 - cos.abs())          # an image without boxes gives NaN, as in the reference
    if fail == len(num_boxes_per_image):
        return None
    return total * 1 / (fail + 1)


def ground_normals(ground_maps, depth_maps, Ks, use_nth=5, id_samples=None, generator=None, plane_cls=None):
    """:1076-1149: per image, back-project every `use_nth`-th pixel of the depth map, keep the ground-mask points (all
    valid points when the image has no ground map, image_size (1,1)), fit a plane with RANSAC (1000 triples, 5 cm) and
    turn its normal towards +y, un-doing a wall hit by a 90 degree turn.  `Ks[i]` is used for image i -- the caller
    passes the per-BOX intrinsics, as in the reference (:1612).  id_samples: optional list of (1000,3) triples."""
    if plane_cls is None:
        from ....ProposalNetwork.utils.plane import Plane as plane_cls
    dev = depth_maps.tensor.device
    out = []
    for i in range(min(len(ground_maps), len(depth_maps), len(Ks))):
        gsize = tuple(ground_maps.image_sizes[i])
        depth = depth_maps[i]
        z = depth[::use_nth, ::use_nth]
        height, width = z.shape
        K = Ks[i]
        fx, fy = torch.floor(K[0, 0] / use_nth), torch.floor(K[1, 1] / use_nth)          # `//` on tensors
        u, v = torch.meshgrid(torch.arange(width, device=dev), torch.arange(height, device=dev), indexing='xy')
        x = (u - width / 2) * z / fx
        y = (v - height / 2) * z / fy
        pts = torch.stack((x, y, z), dim=-1).reshape(-1, 3)
        if gsize != (1, 1):           # only the ground-mask points (their number is data dependent: one host sync)
            pts = pts[torch.nonzero((ground_maps[i][::use_nth, ::use_nth] > 0).reshape(-1)).squeeze(1)]
        best_eq, _ = plane_cls().fit_parallel(pts, thresh=0.05, maxIteration=1000, need_inliers=False,
                                              **({"id_samples": id_samples[i]} if id_samples is not None else {}),
                                              **({"generator": generator} if generator is not None else {}))
        nv = best_eq[:-1]
        # wall instead of floor: swap axes (walls are assumed perpendicular to the floor)
        nv = torch.where(nv[2].abs() > nv[1].abs(), torch.stack((nv[0], nv[2], -nv[1])), nv)
        nv = torch.where(nv[0].abs() > nv[1].abs(), torch.stack((-nv[2], nv[0], nv[1])), nv)
        nv = torch.where(nv[1] < 0, -nv, nv)
        out.append(nv)
    return torch.stack(out)


WEAK_TABLE_COLS = 20


def weak_table(Ks, im_scales_ratio, im_dims, ground_maps, depth_maps, use_nth=5):
    """per-image constants of the weak cube branch as ONE host-built (B,20) float32 tensor (pinned when CUDA is there):
    K / ratio with K[2][2] = 1 (9) | clamp bounds of the projection (4: x with dims[0], y with dims[1], as the reference passes
    them, roi_heads.py:1419-1423,1551) | ground-map confidence (0.1 for an image without ground map, :1600-1606) | height, width
    of the depth map | strided height, width of `ground_normals` | 1 if the image has a ground map | 0."""
    rows = []
    for i in range(len(Ks)):
        k = torch.as_tensor(Ks[i], dtype=torch.float32) / im_scales_ratio[i]
        k[-1, -1] = 1
        d = im_dims[i]
        has_ground = ground_maps is not None and tuple(ground_maps.image_sizes[i]) != (1, 1)
        dh, dw = tuple(depth_maps.image_sizes[i]) if depth_maps is not None else (d[0], d[1])
        rows.append(k.flatten().tolist() + [float(v) for v in _int_clamp_bounds(d[0]) + _int_clamp_bounds(d[1])]
                    + [1.0 if (ground_maps is None or has_ground) else 0.1, float(dh), float(dw), float(-(-int(dh) // use_nth)),
                       float(-(-int(dw) // use_nth)), 1.0 if has_ground else 0.0, 0.0])
    t = torch.tensor(rows, dtype=torch.float32)
    return t.pin_memory() if torch.cuda.is_available() else t


_GRID = {}


def ground_normals_batched(ground_maps, depth_maps, table, kimg=None, use_nth=5, generator=None, n_iter=1000, thresh=0.05):
    """`ground_normals` for every image at once and without a host sync (static shapes: usable on the dense training path):
    the strided back-projection over the padded batch with a validity mask, triples sampled on the device among each image's
    eligible points (Plane.sample_triples_batched), one cr_ransac_plane_batched launch, the axis fix-ups.  table: weak_table()
    on the device; kimg (B) int64: the image whose intrinsics image i is back-projected with (the reference indexes the
    per-BOX intrinsics with the image number, :1612) or None for its own.  An image whose ground map has fewer than three
    pixels set falls back to all its points (the reference would fail there).  -> (B,3)"""
    from .... import geometry as geo
    from ....ProposalNetwork.utils.plane import Plane
    from .boxer import fix_ground_normal
    depth = depth_maps.tensor
    dev, B = depth.device, depth.shape[0]
    zs = depth[:, ::use_nth, ::use_nth]
    Hs, Ws = zs.shape[1], zs.shape[2]
    key = (Hs, Ws, str(dev))
    if key not in _GRID:
        _GRID[key] = (torch.arange(Ws, device=dev, dtype=torch.float32).view(1, 1, Ws),
                      torch.arange(Hs, device=dev, dtype=torch.float32).view(1, Hs, 1))
    u, v = _GRID[key]
    K = table[:, :9] if kimg is None else table[kimg, :9]
    fx, fy = torch.floor(K[:, 0] / use_nth).view(B, 1, 1), torch.floor(K[:, 4] / use_nth).view(B, 1, 1)
    hh, ww = table[:, 16].view(B, 1, 1), table[:, 17].view(B, 1, 1)
    x = (u - ww / 2) * zs / fx
    y = (v - hh / 2) * zs / fy
    inside = ((u < ww) & (v < hh)).reshape(B, -1)
    pts = torch.stack((x, y, zs), dim=-1).reshape(B, -1, 3)
    elig = inside
    if ground_maps is not None:
        g = ground_maps.tensor
        if tuple(g.shape[-2:]) != tuple(depth.shape[-2:]):                  # the two lists are padded separately
            g = torch.nn.functional.pad(g, (0, max(depth.shape[2] - g.shape[2], 0), 0, max(depth.shape[1] - g.shape[1], 0)))
            g = g[:, :depth.shape[1], :depth.shape[2]]
        on = (g[:, ::use_nth, ::use_nth] > 0).reshape(B, -1) | (table[:, 18] == 0).view(B, 1)
        elig = inside & on
        elig = elig | ((elig.sum(1, keepdim=True) < 3) & inside)
    triples = Plane.sample_triples_batched(elig, B, pts.shape[1], n_iter, dev, generator)
    neg_eq, _, _ = geo.ransac_plane_batched(pts, triples, elig, thresh=thresh)
    return fix_ground_normal(neg_eq[:, :3].t()).t().contiguous()


def normal_to_rotation(normal):
    """:1306-1317 (the normalisation by the norm of the WHOLE batch and the `.any() < 0.001` test are the reference's)"""
    n = normal.shape[0]
    x1 = torch.tensor([1.0, 0, 0], device=normal.device).repeat(n, 1)
    t0 = torch.cross(normal, x1, dim=1)
    if torch.bmm(t0.view(n, 1, 3), t0.view(n, 3, 1)).flatten().any() < 0.001:
        y1 = torch.tensor([0, 1.0, 0], device=normal.device).repeat(n, 1)
        t0 = torch.cross(normal, y1, dim=1)
    t0 = t0 / torch.norm(t0)
    t1t = torch.cross(normal, t0, dim=1)
    t1 = t1t / torch.norm(t1t)
    return torch.cat([t0, t1, normal], dim=1).reshape((n, 3, 3))


def z_search_loss(gt_boxes, cubes, K, bounds, proj_boxes, max_count=50):
    """:1151-1194: move each cube along z in 50 steps of 0.1 m (away when its projection is larger than the 2D box,
    closer otherwise), take the step whose projected area is closest to the 2D box's, loss = |z - z_step| / 2
    (a constant w.r.t. the network: both terms carry z); RoIs failing the centre test get 0.1 * 50 / 2.
    gt_boxes, proj_boxes (n,4); cubes (n,15)."""
    n = cubes.shape[0]
    gt_area = (gt_boxes[:, 2] - gt_boxes[:, 0]) * (gt_boxes[:, 3] - gt_boxes[:, 1])
    pc = (proj_boxes[:, :2] + proj_boxes[:, 2:]) / 2
    pred_area = (proj_boxes[:, 2] - proj_boxes[:, 0]) * (proj_boxes[:, 3] - proj_boxes[:, 1])
    # as in the reference: `(a <= c) <= b` compares a boolean with b
    within = ((gt_boxes[:, 0] - max_count <= pc[:, 0]) <= gt_boxes[:, 2] + max_count) & \
             ((gt_boxes[:, 1] - max_count <= pc[:, 1]) <= gt_boxes[:, 3] + max_count)
    values = torch.linspace(0.0, (max_count - 1) / 10, max_count, device=cubes.device)
    sign = torch.where(gt_area < pred_area, 1.0, -1.0).to(cubes.dtype)
    mod = cubes[:, None, :].repeat(1, max_count, 1)
    mod_z = cubes[:, None, 2] + sign[:, None] * values[None, :]
    mod = torch.cat((mod[..., :2], mod_z[..., None], mod[..., 3:]), dim=-1)
    with torch.no_grad():
        boxes = corners_to_boxes(project_cubes_to_corners(mod, K, bounds))
        areas = (boxes[..., 2] - boxes[..., 0]) * (boxes[..., 3] - boxes[..., 1])
        areas = areas + (areas == 0) * 10000000
        idx = torch.argmin((gt_area[:, None] - areas).abs(), dim=1)
    found = (cubes[:, 2] - mod_z.gather(1, idx[:, None])[:, 0]).abs()
    scores = torch.where(within, found, torch.full_like(found, 0.1 * max_count))
    return scores / 2


def _image_index(per_image, device):
    """per-box image index: given as such (int64 tensor) or as the host list of box counts per image"""
    if isinstance(per_image, torch.Tensor):
        return per_image
    return torch.tensor([i for i, num in enumerate(per_image) for _ in range(num)], dtype=torch.int64, device=device)


def pseudo_gt_z_point(depth_maps, xy, num_boxes_per_image):
    """:1256-1279 (target only): depth under the predicted centre, clamped 10 px inside the image."""
    img = _image_index(num_boxes_per_image, xy.device)
    hw = torch.tensor([tuple(s) for s in depth_maps.image_sizes], device=xy.device)[img]
    x = torch.minimum(torch.maximum(xy[:, 0], xy.new_full((), 10.0)), (hw[:, 1] - 11).to(xy.dtype))
    y = torch.minimum(torch.maximum(xy[:, 1], xy.new_full((), 10.0)), (hw[:, 0] - 11).to(xy.dtype))
    return depth_maps.tensor[img, y.long(), x.long()]


def pseudo_gt_z_box(depth_maps, boxes, num_boxes_per_image, median_fn=None):
    """:1196-1232 (target only): median depth inside each projected box clipped to its image (lower median, like
    torch.median); boxes with no area left fall back to the depth under their clamped centre.  As in the reference the
    targets of one image come out as [boxes with area..., boxes without area...], i.e. permuted against the
    predictions when an image has both kinds."""
    dev = boxes.device
    img = _image_index(num_boxes_per_image, dev)
    hw = torch.tensor([tuple(s) for s in depth_maps.image_sizes], device=dev)[img].to(boxes.dtype)
    zero = torch.zeros((), device=dev, dtype=boxes.dtype)
    b = torch.stack((boxes[:, 0].clamp(min=zero).minimum(hw[:, 1]), boxes[:, 1].clamp(min=zero).minimum(hw[:, 0]),
                     boxes[:, 2].clamp(min=zero).minimum(hw[:, 1]), boxes[:, 3].clamp(min=zero).minimum(hw[:, 0])), 1)
    inside = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) > 0
    if median_fn is None:
        from .... import geometry as geo
        median_fn = geo.box_median
    med = median_fn(depth_maps.tensor, b.detach().long().to(torch.int32), img.to(torch.int32))
    ctr = (b[:, :2] + b[:, 2:]) / 2
    x = torch.minimum(torch.maximum(ctr[:, 0], zero + 10.0), hw[:, 1] - 11)
    y = torch.minimum(torch.maximum(ctr[:, 1], zero + 10.0), hw[:, 0] - 11)
    point = depth_maps.tensor[img, y.long(), x.long()]
    target = torch.where(inside, med, point)
    order = torch.sort(img * 2 + (~inside).long(), stable=True).indices
    return target[order]


def dim_hinge_loss(prior_mean, prior_std, dimensions):
    """:1234-1254: how many standard deviations beyond one the predicted (w, h, l) lie from the class prior;
    (None, None, None) when a selected prior has a NaN standard deviation."""
    if bool(torch.isnan(prior_std).any()):
        return None, None, None
    s = ((dimensions - prior_mean).abs() / prior_std - 1.0).clamp(min=0)
    return s[:, 0], s[:, 1], s[:, 2]


# ---------------------------------------------------------------------------------------------- losses that need masks
def mask_index_of(roi_keys, mask_keys):
    """`first_occurrence_indices` of roi_heads.py:866-881 without the host dictionary: the mask of a RoI is the FIRST object
    (over all images, in target order) whose gt_boxes3D[:, 0] equals the RoI's -- objects sharing that value share a
    mask, as in the reference."""
    hit = roi_keys[:, None] == mask_keys[None, :]
    return torch.argmax(hit.to(torch.int32), dim=1)


def segment_loss(masks, bube_corners, mask_idx, hull_fn=None, focal_fn=None):
    """:1030-1053 (loss='focal'): hull of the 8 projected corners -> soft polygon mask -> sigmoid focal loss against the
    object's mask (the mask is the `inputs` argument of the focal loss, as in the reference), mean over the pixels.
    masks (Nm,H,W) uint8; bube_corners (n,8,2) already clamped to the image; mask_idx (n)."""
    if hull_fn is None or focal_fn is None:
        from .... import geometry as geo
        hull_fn, focal_fn = hull_fn or geo.hull8, focal_fn or geo.polygon_focal
    order, count, bump = hull_fn(bube_corners)
    pts = bube_corners + bump[..., None]
    hull = torch.gather(pts, 1, order[..., None].expand(-1, -1, 2))
    return focal_fn(hull, count, masks, mask_idx)


def depth_range_loss(masks, mask_idx, depth_maps, corners_z, gt_boxes, img):
    """:1281-1304: |(q90 - q10 of the depth under the object's mask) - (depth extent of the predicted cuboid)|; the
    depth map is resampled (bilinear, align_corners) to the mask's size when they differ; an empty mask falls back to the
    2D box.  One small sort per RoI, like the reference (torch.quantile)."""
    import torch.nn.functional as F
    pred = corners_z.max(dim=1).values - corners_z.min(dim=1).values
    gts = []
    cache = {}
    for r in range(pred.shape[0]):
        i = int(img[r])
        m = masks[mask_idx[r]].bool()
        if i not in cache:
            d = depth_maps[i]
            cache[i] = d if d.shape == m.shape else F.interpolate(d[None, None], size=m.shape, mode='bilinear', align_corners=True)[0, 0]
        d = cache[i]
        vals = d[m]
        if vals.numel() == 0:
            b = gt_boxes[r].long()
            vals = d[b[1]:b[3], b[0]:b[2]]
        gts.append(torch.quantile(vals.flatten(), 0.9) - torch.quantile(vals.flatten(), 0.1))
    return (torch.stack(gts) - pred).abs()
