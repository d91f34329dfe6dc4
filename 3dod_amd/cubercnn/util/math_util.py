"""3D math of the Cube R-CNN head, differentiable torch versions (small per-RoI tensors; the
fused device kernels live in 3dod_amd.geometry).  Restates cubercnn/util/math_util.py of the
reference; the pytorch3d transforms it imports are restated from their published definition
[third-party: pytorch3d.transforms.rotation_6d_to_matrix / axis_angle_to_matrix]."""
import math
from collections import defaultdict

import torch
import torch.nn.functional as F


def to_float_tensor(x):
    if not isinstance(x, torch.Tensor):
        x = torch.tensor(x)
    return x.float()


_SX = (-1, 1, 1, -1, -1, 1, 1, -1)     # x <- l : -l/2 for {0,3,4,7}
_SY = (-1, -1, 1, 1, -1, -1, 1, 1)     # y <- h : -h/2 for {0,1,4,5}
_SZ = (-1, -1, -1, -1, 1, 1, 1, 1)     # z <- w : -w/2 for {0,1,2,3}


_CUBOID_CONST = {}


def _cuboid_constants(dev):
    """corner sign table and triangle list, made once per device (no host-to-device copy per call)"""
    c = _CUBOID_CONST.get(str(dev))
    if c is None:
        signs = torch.tensor([_SX, _SY, _SZ], dtype=torch.float32, device=dev)
        faces = torch.tensor([[0, 1, 2], [2, 3, 0], [1, 5, 6], [6, 2, 1], [4, 0, 3], [3, 7, 4], [5, 4, 7], [7, 6, 5],
                              [4, 5, 1], [1, 0, 4], [3, 2, 6], [6, 7, 3]], device=dev).float()
        c = _CUBOID_CONST[str(dev)] = (signs, faces)
    return c


def get_cuboid_verts_faces(box3d=None, R=None):
    """math_util.py:142-245.  box3d (n,6) [X,Y,Z,W,H,L], R (n,3,3) -> verts (n,8,3), faces (n,12,3)."""
    if box3d is None:
        box3d = [0, 0, 0, 1, 1, 1]
    box3d = to_float_tensor(box3d)
    if R is not None:
        R = to_float_tensor(R)
    squeeze = box3d.dim() == 1
    if squeeze:
        box3d = box3d.unsqueeze(0)
        if R is not None:
            R = R.unsqueeze(0)
    n = len(box3d)
    signs, faces0 = _cuboid_constants(box3d.device)                       # (3,8) half-extent signs for (l, h, w); (12,3)
    half = torch.stack((box3d[:, 5], box3d[:, 4], box3d[:, 3]), dim=1) * 0.5
    verts = signs[None] * half[:, :, None]                                # (n,3,8)
    if R is not None:
        verts = R @ verts
    verts = verts + box3d[:, :3].unsqueeze(2)
    verts = verts.transpose(1, 2)
    faces = faces0.unsqueeze(0).expand(n, 12, 3)
    if squeeze:
        verts = verts.squeeze()
        faces = faces.squeeze()
    return verts, faces


def mat2euler(R):
    """math_util.py:71-81: XYZ Euler angles of a rotation matrix (numpy array of 3)"""
    import math
    import numpy as np
    sy = math.sqrt(R[0, 0] * R[0, 0] + R[1, 0] * R[1, 0])
    return np.array([math.atan2(R[2, 1], R[2, 2]), math.atan2(-R[2, 0], sy), math.atan2(R[1, 0], R[0, 0])])


def compute_virtual_scale_from_focal_spaces(f, H, f0, H0):
    """math_util.py:732-743."""
    return (H0 * f) / (f0 * H)


def rotation_6d_to_matrix(d6):
    """pytorch3d.transforms.rotation_6d_to_matrix [third-party, restated]: Gram-Schmidt, rows (b1,b2,b3)."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = F.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def quaternion_to_matrix(q):
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def axis_angle_to_matrix(axis_angle):
    """pytorch3d.transforms.axis_angle_to_matrix [third-party, restated] (via quaternions)."""
    angles = torch.norm(axis_angle, p=2, dim=-1, keepdim=True)
    half = angles * 0.5
    eps = 1e-6
    small = angles.abs() < eps
    s = torch.empty_like(angles)
    s[~small] = torch.sin(half[~small]) / angles[~small]
    s[small] = 0.5 - (angles[small] * angles[small]) / 48
    quat = torch.cat([torch.cos(half), axis_angle * s], dim=-1)
    return quaternion_to_matrix(quat)


def _ray_rotation(K, u, v):
    fx, fy, sx, sy = K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]
    oray = torch.stack(((u - sx) / fx, (v - sy) / fy, torch.ones_like(u))).T
    oray = oray / torch.linalg.norm(oray, dim=1).unsqueeze(1)
    angle = torch.acos(oray[:, -1])
    axis = torch.zeros_like(oray)
    axis[:, 0] = axis[:, 0] - oray[:, 1]
    axis[:, 1] = axis[:, 1] + oray[:, 0]
    norms = torch.linalg.norm(axis, dim=1)
    valid = angle > 0
    M = axis_angle_to_matrix(angle.unsqueeze(1) * axis / norms.unsqueeze(1))
    return M, valid


def R_from_allocentric(K, R_view, u=None, v=None):
    """math_util.py:802-830 (tensor branch): R = M @ R_view where the viewing-ray angle > 0."""
    M, valid = _ray_rotation(K, u, v)
    # branch-free (boolean indexing costs a host sync per use): rows on the optical axis keep R_view; their M (0/0) is
    # replaced by the identity before the product so that no NaN reaches the gradient of the unselected branch
    sel = valid[:, None, None]
    M = torch.where(sel, M, torch.eye(3, dtype=M.dtype, device=M.device).expand_as(M))
    return torch.where(sel, torch.bmm(M, R_view), R_view)


def R_to_allocentric(K, R, u=None, v=None):
    """math_util.py:746-776 (tensor branch)."""
    M, valid = _ray_rotation(K, u, v)
    sel = valid[:, None, None]
    M = torch.where(sel, M, torch.eye(3, dtype=M.dtype, device=M.device).expand_as(M))
    return torch.where(sel, torch.bmm(M.transpose(2, 1), R), R)


def scaled_sigmoid(vals, min=0.0, max=1.0):
    return min + (max - min) * torch.sigmoid(vals)


def cluster_depth(cube_z, box_classes, src_boxes, priors_z_scales, z_type="direct", priors_z_stats=None):
    """the RoI's raw depth out of the (n, bins, K, 1) predictor output and its decode (roi_heads.py:2343-2356, 2404-2436; the
    weak head has the same lines at :1436-1449, :1495-1530): the bin of a RoI is the one whose 2D-scale centre is closest to
    the diagonal of its proposal box, per category; Z_TYPE 'sigmoid' / 'log' / 'clusters' before the virtual-depth factor.
    cube_z (n,bins,K,1) for bins > 1 or (n,K,1); -> (n,)"""
    n = cube_z.shape[0]
    fg = torch.arange(n, device=cube_z.device)
    assign = None
    if cube_z.dim() == 4:
        scales = ((src_boxes[:, 3] - src_boxes[:, 1]) ** 2 + (src_boxes[:, 2] - src_boxes[:, 0]) ** 2).sqrt()
        diff = (priors_z_scales.detach().T.unsqueeze(0) - scales.unsqueeze(1).unsqueeze(2)).abs()        # (n, bins, K)
        assign = diff.argmin(1)[fg, box_classes]
        z = cube_z[fg, assign, box_classes, 0]
    else:
        z = cube_z[fg, box_classes, 0]
    if z_type == 'sigmoid':
        z = torch.sigmoid(z) * 100
    elif z_type == 'log':
        z = torch.exp(z)
    elif z_type == 'clusters':
        st = priors_z_stats.detach()[box_classes, assign]                                             # (n, 2)
        z = scaled_sigmoid(z, min=(st[:, 0] - 3 * st[:, 1]).clip(0), max=st[:, 0] + 3 * st[:, 1])
    elif z_type != 'direct':
        raise ValueError(f"Z_TYPE '{z_type}'")
    return z


def approx_eval_resolution(h, w, scale_min=0, scale_max=1e10):
    """math_util.py:288-316: resolution an h x w image is evaluated at (shortest edge -> scale_min, then capped so the
    longest edge <= scale_max); returns (h, w, factor original -> network)."""
    h0 = h
    s = scale_min / min(h, w)
    h, w = h * s, w * s
    s = min(scale_max / max(h, w), 1.0)
    h, w = h * s, w * s
    return h, w, h / h0


def _mean_std(x):
    """pandas semantics: sample standard deviation (ddof = 1), NaN when undefined."""
    import numpy as np
    x = np.asarray(x, dtype=np.float64)
    mean = float(x.mean()) if x.size else float('nan')
    std = float(x.std(ddof=1)) if x.size > 1 else float('nan')
    return [mean, std]


def compute_priors(cfg, datasets, max_cluster_rounds=1000, min_points_for_std=5, n_bins=None, category_names=None):
    """math_util.py:318-524: statistics of the training annotations the cube head starts from: per-category mean/std
    of the 3D dimensions, of depth z (in VIRTUAL depth when MODEL.ROI_CUBE_HEAD.VIRTUAL_DEPTH) and of y, global z / y
    statistics, and -- for CLUSTER_BINS > 1 -- depth statistics per 2D-scale cluster (1-D k-means on the box diagonal at
    test resolution).  `datasets` is an `Omni3D` index; `category_names` defaults to the model's thing_classes."""
    import numpy as np
    from ...d2lite.data import BoxMode, MetadataCatalog
    anns = datasets.loadAnns(datasets.getAnnIds())
    if category_names is None:
        category_names = MetadataCatalog.get('omni3d_model').thing_classes
    H = cfg.MODEL.ROI_CUBE_HEAD
    smin, smax = cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST

    rows = defaultdict(list)            # category -> [w3d, h3d, l3d, y3d, z3d, scale]
    all_z, all_y = [], []
    for ann in anns:
        name = ann['category_name'].lower()
        img = datasets.imgs[ann['image_id']]
        fy, im_h, im_w = img['K'][1][1], img['height'], img['width']
        if cfg.DATASETS.MODAL_2D_BOXES and 'bbox2D_tight' in ann and ann['bbox2D_tight'][0] != -1:
            box = ann['bbox2D_tight']
        elif cfg.DATASETS.TRUNC_2D_BOXES and 'bbox2D_trunc' in ann and not all(v == -1 for v in ann['bbox2D_trunc']):
            box = ann['bbox2D_trunc']
        elif 'bbox2D_proj' in ann:
            box = ann['bbox2D_proj']
        else:
            continue
        _, _, w, h = BoxMode.convert(box, BoxMode.XYXY_ABS, BoxMode.XYWH_ABS)
        _, y3d, z3d = ann['center_cam']
        w3d, h3d, l3d = ann['dimensions']
        test_h, _, sf = approx_eval_resolution(im_h, im_w, smin, smax)
        h, w = h * sf, w * sf
        if H.VIRTUAL_DEPTH:
            z3d = z3d * (1 / compute_virtual_scale_from_focal_spaces(fy, im_h, H.VIRTUAL_FOCAL, test_h))
        if (not ann['ignore']) and name in category_names:
            rows[name].append([w3d, h3d, l3d, y3d, z3d, math.sqrt(h ** 2 + w ** 2)])
            all_z.append(z3d)
            all_y.append(y3d)

    if n_bins is None:
        n_bins = H.CLUSTER_BINS
    dims_per_cat, z_per_cat, y_per_cat, bins = [], [], [], []
    for cat in category_names:
        d = np.asarray(rows.get(cat, []), dtype=np.float64).reshape(-1, 6)
        n = len(d)
        if n > 0:
            ms = [_mean_std(d[:, i]) for i in range(3)]
            dims_per_cat.append([[m[0] for m in ms], [m[1] for m in ms]])
            z_per_cat.append(_mean_std(d[:, 4]))
            y_per_cat.append(_mean_std(d[:, 3]))
        else:                                            # placeholder statistics for a category without samples
            dims_per_cat.append([[1.0, 1.0, 1.0], [1.0, 1.0, 1.0]])
            z_per_cat.append([50, 50])
            y_per_cat.append([1, 10])
        if n_bins <= 1:
            continue
        if n < min_points_for_std:
            print('Warning {} category has only {} valid samples...'.format(cat, n))
            lo, hi = cfg.MODEL.ANCHOR_GENERATOR.SIZES[0][0], cfg.MODEL.ANCHOR_GENERATOR.SIZES[-1][-1]
            base = (hi / lo) ** (1 / (n_bins - 1))
            centres = [lo * (base ** i) for i in range(n_bins)]
            z_bins = [[b, 15] for b in np.arange(100, 1, -(100 - 1) / n_bins)]
            assert len(z_bins) == n_bins, 'Broken default bin scaling.'
            bins.append((cat, centres, z_bins))
            continue
        scales = torch.tensor(d[:, 5], dtype=torch.float32)

        def members(assign, quality, b):
            m = assign == b
            if m.sum() < min_points_for_std:             # thin cluster: borrow its nearest points
                m[quality[:, b].topk(min_points_for_std)[1]] = True
            return m

        base = (scales.max() / scales.min()) ** (1 / (n_bins - 1))
        centres = torch.tensor([float(scales.min() * (base ** i)) for i in range(n_bins)], dtype=torch.float32)
        best = -np.inf
        for _ in range(max_cluster_rounds):
            quality = -(centres.unsqueeze(0) - scales.unsqueeze(1)).abs()
            score, assign_round = quality.max(1)
            round_score = score.mean().item()
            if np.round(round_score, 5) > best:
                best, assign = round_score, assign_round
                centres = torch.tensor([scales[members(assign, quality, b)].mean().item() for b in range(n_bins)],
                                       dtype=torch.float32)
            else:
                break
        z_bins = [_mean_std(d[members(assign, quality, b).numpy(), 4]) for b in range(n_bins)]
        bins.append((cat, centres.numpy().tolist(), z_bins))

    return {'priors_dims_per_cat': dims_per_cat, 'priors_z3d_per_cat': z_per_cat, 'priors_y3d_per_cat': y_per_cat,
            'priors_bins': bins, 'priors_y3d': _mean_std(all_y), 'priors_z3d': _mean_std(all_z)}
