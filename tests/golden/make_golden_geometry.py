"""Generates tests/golden/geometry_*.npz from the REFERENCE implementation
(imported from /root/reference with stubbed third-party modules, see
_refimport.py).  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_geometry.py

The fixtures hold inputs + the reference's outputs (data only).
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402

ref = _refimport.load()
torch.manual_seed(0)
np.random.seed(0)
torch.set_num_threads(1)


def rand_rot(n, gen):
    q = torch.randn(n, 4, generator=gen)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)
    return R.view(n, 3, 3)


def g1_corners():
    g = torch.Generator().manual_seed(1)
    n = 256
    box = torch.cat([torch.randn(n, 3, generator=g) * 3, torch.rand(n, 3, generator=g) * 2 + 0.05], 1)
    R = rand_rot(n, g)
    verts, faces = ref.math_util.get_cuboid_verts_faces(box, R)
    np.savez_compressed(os.path.join(HERE, "geometry_g1_corners.npz"),
                        box6=box.numpy(), R=R.numpy(), verts=verts.numpy(), faces=faces[0].numpy(),
                        notes="cubercnn/util/math_util.py:142-245 get_cuboid_verts_faces; pinned by reference")


def make_cubes(N, P, g, K, W, H):
    """plausible + adversarial cubes."""
    z = torch.rand(N, P, generator=g) * 6 + 1.0
    u = torch.rand(N, P, generator=g) * W
    v = torch.rand(N, P, generator=g) * H
    x = (u - K[0, 2]) * z / K[0, 0]
    y = (v - K[1, 2]) * z / K[1, 1]
    dims = torch.rand(N, P, 3, generator=g) * 1.5 + 0.05
    R = rand_rot(N * P, g).view(N, P, 9)
    cubes = torch.cat([x[..., None], y[..., None], z[..., None], dims, R], 2)
    # adversarial: very near / behind camera, far off-screen
    cubes[:, 0:20, 2] = torch.rand(N, 20, generator=g) * 0.2 + 0.01
    cubes[:, 20:30, 2] = -torch.rand(N, 10, generator=g) * 2
    cubes[:, 30:50, 0] = (torch.rand(N, 20, generator=g) - 0.5) * 100
    cubes[:, 50:60, 1] = (torch.rand(N, 10, generator=g) - 0.5) * 100
    return cubes.float()


def g2_g3_project_score():
    g = torch.Generator().manual_seed(2)
    N, P, W, H = 4, 1000, 512, 512
    K = torch.tensor([[512., 0, 256], [0, 512., 256], [0, 0, 1]])
    cubes_t = make_cubes(N, P, g, K, W, H)
    cubes = ref.spaces.Cubes(cubes_t)
    im_shape = (W, H)
    corners3d = cubes.get_all_corners()
    bube = cubes.get_bube_corners(K, im_shape)
    boxes = ref.conversions.cubes_to_box(cubes, K, im_shape)
    boxes_t = torch.stack([b.tensor for b in boxes])
    # reference 2D boxes and priors per object
    ctr = torch.rand(N, 2, generator=g) * 300 + 100
    wh = torch.rand(N, 2, generator=g) * 150 + 40
    ref_boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    mu = torch.rand(N, 3, generator=g) * 0.8 + 0.3
    sg = 0.2 * mu
    rect = torch.stack([torch.stack([ref_boxes[i, [0, 1]], ref_boxes[i, [2, 1]],
                                     ref_boxes[i, [2, 3]], ref_boxes[i, [0, 3]]]) for i in range(N)])
    # rotate the rect a little so it is a general minAreaRect-like quad
    ang = torch.rand(N, generator=g) * 0.6 - 0.3
    c = rect.mean(1, keepdim=True)
    rot = torch.stack([torch.stack([ang.cos(), -ang.sin()], 1), torch.stack([ang.sin(), ang.cos()], 1)], 1)
    rect = ((rect - c) @ rot.transpose(1, 2) + c).float()
    iou = torch.zeros(N, P); dim = torch.zeros(N, P); cor = torch.zeros(N, P)
    chamfer = torch.zeros(N, P)
    comb = np.zeros((N, P), np.float32); amax = np.zeros(N, np.int64)
    for i in range(N):
        gt_box = ref.Boxes(ref_boxes[i:i + 1])
        iou[i] = ref.scorefunction.score_iou(gt_box, boxes[i])
        dim[i] = ref.scorefunction.score_dimensions((mu[i], sg[i]), cubes[i].dimensions[0], gt_box, boxes[i])
        # score_corners arithmetic with a GIVEN rect (cv2 part replaced by an input),
        # scorefunction.py:76-85 restated around the reference's own modified_chamfer_distance
        bc = bube[i]
        sc = torch.zeros(P)
        for j in range(P):
            sc[j] = ref.scorefunction.modified_chamfer_distance(rect[i].numpy(), bc[j].numpy())
        chamfer[i] = sc
        cor[i] = 1 - sc / torch.max(sc)
        comb[i] = np.array(iou[i]) * np.array(dim[i]) * np.array(cor[i])     # roi_heads.py:499
        amax[i] = np.argmax(comb[i])
    np.savez_compressed(os.path.join(HERE, "geometry_g2_project_score.npz"),
                        cubes=cubes_t.numpy(), K=K.numpy(), im_wh=np.array(im_shape, np.int64),
                        corners3d=corners3d.numpy(), corners2d=bube.numpy(), boxes=boxes_t.numpy(),
                        ref_boxes=ref_boxes.numpy(), prior_mu=mu.numpy(), prior_sigma=sg.numpy(),
                        rect_pts=rect.numpy(), iou=iou.numpy(), dim=dim.numpy(), chamfer=chamfer.numpy(),
                        corner=cor.numpy(), combined=comb, argmax=amax,
                        notes="spaces.py:192-245, conversions.py:25-48, scorefunction.py:47-85,144-160, "
                              "roi_heads.py:492-505. corners/boxes/dim pinned by reference; iou flows through a "
                              "pairwise_iou stand-in (detectron2 absent: parity unpinned); corner score uses the "
                              "reference's modified_chamfer_distance (scipy) with a given rect")


def g5_yaw_table():
    normals = torch.tensor([[0., 1., 0.], [0.1, 0.98, -0.17], [-0.3, 0.9, 0.3]])
    normals = normals / normals.norm(dim=1, keepdim=True)
    angles = torch.linspace(0, np.pi, 36)
    tabs = torch.stack([ref.utils.orthobasis_from_normal_t(n, angles) for n in normals])
    np.savez_compressed(os.path.join(HERE, "geometry_g5_yaw_table.npz"),
                        normals=normals.numpy(), angles=angles.numpy(), tables=tabs.numpy(),
                        notes="ProposalNetwork/utils/utils.py:112-146; pinned by reference")


def g_propose():
    """propose() with torch.normal / torch.randint wrapped to RECORD the variates
    (standard normals and indices) they consume, proposals.py:338-424."""
    g = torch.Generator().manual_seed(5)
    N, P, W, H = 6, 1000, 512, 512
    K = torch.tensor([[600., 0, 250], [0, 600., 260], [0, 0, 1]])
    depth = (torch.rand(H, W, generator=g) * 3 + 1).float()
    ctr = torch.rand(N, 2, generator=g) * 300 + 100
    wh = torch.rand(N, 2, generator=g) * 150 + 40
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1).clamp(0, 511)
    mu = torch.rand(N, 3, generator=g) * 0.8 + 0.3
    sg = 0.35 * mu                     # wide enough that the rejection loop runs
    normal = torch.tensor([0.05, 0.99, -0.1]); normal = normal / normal.norm()
    rec = {"normals": [], "randint": []}
    real_normal, real_randint = torch.normal, torch.randint

    def normal_rec(mean, std, *a, **k):
        n = torch.randn(mean.shape, generator=g)
        rec["normals"].append(n.clone())
        return mean + std * n

    def randint_rec(high, size, *a, **k):
        r = real_randint(high, size, generator=g)
        rec["randint"].append(r.clone())
        return r

    torch.normal, torch.randint = normal_rec, randint_rec
    try:
        cubes, _, _ = ref.proposals.propose(ref.Boxes(boxes), depth, (mu, sg), (W, H), K,
                                            number_of_proposals=P, ground_normal=normal)
    finally:
        torch.normal, torch.randint = real_normal, real_randint
    # order of normal draws in propose: w rounds..., h rounds..., l rounds..., x, y, z
    # sample_normal_in_range draws once, then once per rejection round.
    np.savez_compressed(os.path.join(HERE, "geometry_g6_propose.npz"),
                        boxes=boxes.numpy(), depth=depth.numpy(), prior_mu=mu.numpy(), prior_sigma=sg.numpy(),
                        K=K.numpy(), normal=normal.numpy(), P=np.int64(P),
                        normals=np.stack([n.numpy() for n in rec["normals"]]),
                        yaw_idx=rec["randint"][0].numpy(), cubes=cubes.tensor.numpy(),
                        notes="proposals.py:338-424 with recorded torch.normal/randint variates; the fixture's "
                              "`normals` (D,N,P) are in draw order: w + its rejection rounds, h + rounds, "
                              "l + rounds, x, y, z. pinned by reference")


def g8_ransac():
    g = torch.Generator().manual_seed(8)
    Q, T = 4000, 1000
    xy = (torch.rand(Q, 2, generator=g) - 0.5) * 6
    yplane = 1.4 + 0.02 * torch.randn(Q, generator=g) + 0.05 * xy[:, 0]
    pts = torch.stack([xy[:, 0], yplane, xy[:, 1] + 4], 1)
    pts[:800] = torch.rand(800, 3, generator=g) * 4     # outliers
    rnd = random.Random(8)
    triples = [rnd.sample(range(0, Q), 3) for _ in range(T)]
    it = iter(triples)
    real = ref.plane.random.sample
    ref.plane.random.sample = lambda pop, k: next(it)
    try:
        eq, inl = ref.plane.Plane().fit_parallel(pts, thresh=0.05, maxIteration=T)
    finally:
        ref.plane.random.sample = real
    np.savez_compressed(os.path.join(HERE, "geometry_g8_ransac.npz"),
                        pts=pts.numpy(), triples=np.array(triples, np.int32), thresh=np.float32(0.05),
                        neg_equation=eq.numpy(), n_inliers=np.int64(len(inl)), inliers=inl.numpy(),
                        notes="ProposalNetwork/utils/plane.py:79-134 with given triples; pinned by reference")


if __name__ == "__main__":
    g1_corners()
    g2_g3_project_score()
    g5_yaw_table()
    g_propose()
    g8_ransac()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
