"""3D math of the Cube R-CNN head, differentiable torch versions (small per-RoI tensors; the
fused device kernels live in 3dod_amd.geometry).  Restates cubercnn/util/math_util.py of the
reference; the pytorch3d transforms it imports are restated from their published definition
[third-party: pytorch3d.transforms.rotation_6d_to_matrix / axis_angle_to_matrix]."""
import torch
import torch.nn.functional as F


def to_float_tensor(x):
    if not isinstance(x, torch.Tensor):
        x = torch.tensor(x)
    return x.float()


_SX = (-1, 1, 1, -1, -1, 1, 1, -1)     # x <- l : -l/2 for {0,3,4,7}
_SY = (-1, -1, 1, 1, -1, -1, 1, 1)     # y <- h : -h/2 for {0,1,4,5}
_SZ = (-1, -1, -1, -1, 1, 1, 1, 1)     # z <- w : -w/2 for {0,1,2,3}


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
    dev = box3d.device
    sx = torch.tensor(_SX, dtype=torch.float32, device=dev)
    sy = torch.tensor(_SY, dtype=torch.float32, device=dev)
    sz = torch.tensor(_SZ, dtype=torch.float32, device=dev)
    verts = torch.stack((sx[None] * (box3d[:, 5:6] / 2), sy[None] * (box3d[:, 4:5] / 2),
                         sz[None] * (box3d[:, 3:4] / 2)), dim=1)          # (n,3,8)
    if R is not None:
        verts = R @ verts
    verts = verts + box3d[:, :3].unsqueeze(2)
    verts = verts.transpose(1, 2)
    faces = torch.tensor([[0, 1, 2], [2, 3, 0], [1, 5, 6], [6, 2, 1], [4, 0, 3], [3, 7, 4], [5, 4, 7], [7, 6, 5],
                          [4, 5, 1], [1, 0, 4], [3, 2, 6], [6, 7, 3]], device=dev).float().unsqueeze(0).repeat([n, 1, 1])
    if squeeze:
        verts = verts.squeeze()
        faces = faces.squeeze()
    return verts, faces


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
    R = R_view.clone()
    R[valid] = torch.bmm(M[valid], R_view[valid])
    return R


def R_to_allocentric(K, R, u=None, v=None):
    """math_util.py:746-776 (tensor branch)."""
    M, valid = _ray_rotation(K, u, v)
    R_view = R.clone()
    R_view[valid] = torch.bmm(M[valid].transpose(2, 1), R[valid])
    return R_view


def scaled_sigmoid(vals, min=0.0, max=1.0):
    return min + (max - min) * torch.sigmoid(vals)
