"""Drawing helpers of the demo / evaluation visualisations (reference: cubercnn/vis/vis.py), on PIL instead of cv2 and
without the pytorch3d mesh renderer: 3D boxes are drawn as projected wireframes (clipped at a near plane), the "novel
view" of `draw_scene_view` is an orthographic top-down plot of the box footprints.  Images are HxWx3 uint8 arrays in the
caller's channel order (BGR in tools/demo.py, like cv2); drawing functions modify `im` in place and return it.
[not part of the accelerated path; the rendered pixels are not compared with the reference's]"""
import numpy as np
from PIL import Image, ImageDraw, ImageFont

__all__ = ["interp_color", "draw_line", "draw_2d_box", "draw_circle", "draw_text", "draw_3d_box_from_verts", "draw_3d_box",
           "draw_bev", "draw_scene_view", "imhstack", "imvstack", "CUBOID_EDGES"]

# edges of the 8 corners as cubercnn.util.get_cuboid_verts_faces orders them (two quads 0-1-2-3 / 4-5-6-7 + connectors)
CUBOID_EDGES = ((0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7))
_FONT = [None]


def _font(size):
    try:
        return ImageFont.load_default(size=size)
    except TypeError:                       # older Pillow: fixed-size bitmap font
        if _FONT[0] is None:
            _FONT[0] = ImageFont.load_default()
        return _FONT[0]


def _draw_on(im, fn):
    """run fn(ImageDraw) on a PIL view of `im` and copy the pixels back (PIL cannot draw into a numpy buffer)"""
    pil = Image.fromarray(im)
    fn(ImageDraw.Draw(pil))
    im[...] = np.asarray(pil)
    return im


def interp_color(dist, bounds=(0, 1), color_lo=(0, 0, 250), color_hi=(0, 250, 250)):
    """vis.py:17-24: linear blend of two colours by where `dist` lies in `bounds`"""
    t = min(max((dist - bounds[0]) / (bounds[1] - bounds[0]), 0.0), 1.0)
    return tuple(float(a) * (1 - t) + float(b) * t for a, b in zip(color_lo, color_hi))


def _rgb(color):
    return tuple(int(round(c)) for c in color[:3])


def draw_line(im, v0, v1, color=(0, 200, 200), thickness=1):
    return _draw_on(im, lambda d: d.line([(float(v0[0]), float(v0[1])), (float(v1[0]), float(v1[1]))], fill=_rgb(color),
                                         width=int(thickness)))


def draw_2d_box(im, box, color=(0, 200, 200), thickness=1):
    """box = [x, y, w, h] as in the reference (vis.py:712-722)"""
    x, y, w, h = [float(v) for v in box]
    return _draw_on(im, lambda d: d.rectangle([x, y, x + w, y + h], outline=_rgb(color), width=int(thickness)))


def draw_circle(im, pos, radius=5, thickness=1, color=(250, 100, 100), fill=True):
    x, y = float(pos[0]), float(pos[1])
    return _draw_on(im, lambda d: d.ellipse([x - radius, y - radius, x + radius, y + radius], outline=_rgb(color),
                                            fill=_rgb(color) if fill else None, width=int(thickness)))


def draw_text(im, text, pos, scale=0.4, color='auto', bg_color=(0, 255, 255), blend=0.33, lineType=1):
    """label with a filled background; `scale` follows cv2's font scale (0.4 ~ 11 px)"""
    font = _font(max(8, int(round(scale * 28))))
    x, y = int(pos[0]), int(pos[1])
    if color == 'auto':
        color = (0, 0, 0) if sum(bg_color) / 3 > 127 else (255, 255, 255)

    def fn(d):
        l, t, r, b = d.textbbox((x, y), text, font=font)
        if bg_color is not None:
            d.rectangle([l - 1, t - 1, r + 1, b + 1], fill=_rgb(bg_color))
        d.text((x, y), text, fill=_rgb(color), font=font)
    return _draw_on(im, fn)


def _project(K, p):
    return (K[0][0] * p[0] / p[2] + K[0][2], K[1][1] * p[1] / p[2] + K[1][2])


def draw_3d_box_from_verts(im, K, verts3d, color=(0, 200, 200), thickness=1, draw_back=False, draw_top=False, zplane=0.05,
                           eps=1e-4):
    """wireframe of 8 camera-space corners; edges are clipped at z = zplane instead of dropped (vis.py:578-653)"""
    K = np.asarray(K, dtype=np.float64)
    v = np.asarray(verts3d, dtype=np.float64).reshape(8, 3)
    segs = []
    for a, b in CUBOID_EDGES:
        p, q = v[a], v[b]
        if p[2] < zplane and q[2] < zplane:
            continue
        if p[2] < zplane or q[2] < zplane:
            t = (zplane - p[2]) / (q[2] - p[2] + (eps if q[2] == p[2] else 0.0))
            cut = p + t * (q - p)
            p, q = (cut, q) if p[2] < zplane else (p, cut)
        segs.append((_project(K, p), _project(K, q)))

    def fn(d):
        for s0, s1 in segs:
            d.line([s0, s1], fill=_rgb(color), width=int(thickness))
    return _draw_on(im, fn)


def draw_3d_box(im, K, box3d, R, color=(0, 200, 200), thickness=1, draw_back=False, draw_top=False, view_R=None, view_T=None):
    """box3d = [X, Y, Z, W, H, L] in camera space, R its 3x3 pose (vis.py:655-658)"""
    import torch
    from ..util import math_util as util
    verts = util.get_cuboid_verts_faces(torch.as_tensor(box3d, dtype=torch.float32), torch.as_tensor(R, dtype=torch.float32))[0]
    verts = verts.cpu().numpy().astype(np.float64)
    if view_R is not None:
        verts = verts @ np.asarray(view_R, dtype=np.float64).T
    if view_T is not None:
        verts = verts + np.asarray(view_T, dtype=np.float64)
    return draw_3d_box_from_verts(im, K, verts, color=color, thickness=thickness, draw_back=draw_back, draw_top=draw_top)


def draw_bev(canvas_bev, z3d, l3d, w3d, x3d, ry3d, color=(0, 200, 200), scale=1, thickness=2):
    """one rotated footprint on a bird's-eye canvas whose bottom centre is the camera (vis.py:26-56); units * scale = px"""
    h, w = canvas_bev.shape[:2]
    c, s = np.cos(ry3d), np.sin(ry3d)
    corners = np.array([[l3d / 2, w3d / 2], [l3d / 2, -w3d / 2], [-l3d / 2, -w3d / 2], [-l3d / 2, w3d / 2]])
    pts = corners @ np.array([[c, -s], [s, c]]).T + [x3d, z3d]
    px = [(w / 2 + p[0] * scale, h - p[1] * scale) for p in pts]
    return _draw_on(canvas_bev, lambda d: d.line(px + [px[0]], fill=_rgb(color), width=int(thickness)))


def draw_scene_view(im, K, boxes, text=None, scale=1000, colors=None, thickness=2, blend_weight=0.80, blend_weight_overlay=1.0,
                    ground_bounds=None, canvas=None, zplane=0.05, **unused):
    """front view (wireframes + labels over `im`) and a top-down view of the same boxes.  `boxes`: list of
    (bbox3D [X,Y,Z,W,H,L], pose 3x3) or of objects with .bbox3D / .pose / .color (see util.mesh_cuboid).  Returns
    (im_front, im_topdown, canvas) like the reference (vis.py:210-545), whose novel view is a rendered mesh scene."""
    import torch
    from ..util import math_util as util
    front = np.ascontiguousarray(im).copy()
    items = []
    for i, b in enumerate(boxes):
        bbox3D, pose = (b.bbox3D, b.pose) if hasattr(b, "bbox3D") else b
        col = colors[i] if colors is not None else getattr(b, "color", None)
        if col is None:
            from ..util.util import get_color
            col = get_color(i)
        if max(col) <= 1.0:
            col = [c * 255.0 for c in col]
        verts = util.get_cuboid_verts_faces(torch.as_tensor(bbox3D, dtype=torch.float32),
                                            torch.as_tensor(pose, dtype=torch.float32))[0].cpu().numpy().astype(np.float64)
        items.append((verts, _rgb(col)))
    overlay = front.copy()
    for verts, col in items:
        draw_3d_box_from_verts(overlay, K, verts, color=col, thickness=thickness, zplane=zplane)
    front[...] = (blend_weight_overlay * overlay + (1 - blend_weight_overlay) * front).astype(np.uint8)
    if text is not None:
        for (verts, col), t in zip(items, text):
            vis_pts = verts[verts[:, 2] > zplane]
            if len(vis_pts):
                uv = np.array([_project(np.asarray(K, dtype=np.float64), p) for p in vis_pts])
                draw_text(front, t, (float(uv[:, 0].min()), max(float(uv[:, 1].min()) - 12, 0)), bg_color=col)
    # top-down: X to the right, Z up the canvas, camera at the bottom centre
    size = int(scale)
    top = np.full((size, size, 3), 255, dtype=np.uint8) if canvas is None else canvas
    if items:
        allv = np.concatenate([v for v, _ in items])
        if ground_bounds is None:
            zmax = max(float(allv[:, 2].max()) * 1.1, 1.0)
            xmax = max(float(np.abs(allv[:, 0]).max()) * 1.1, zmax / 2)
        else:
            xmax, zmax = ground_bounds
        s = min(size / (2 * xmax), size / zmax)

        def fn(d):
            for verts, col in items:
                foot = verts[[0, 1, 5, 4]] if np.ptp(verts[[0, 1, 5, 4], 1]) < np.ptp(verts[[0, 1, 2, 3], 1]) else verts[[0, 1, 2, 3]]
                px = [(size / 2 + p[0] * s, size - p[2] * s) for p in foot]
                d.polygon(px, outline=col)
                d.line(px + [px[0]], fill=col, width=int(thickness))
            d.polygon([(size / 2 - 6, size - 1), (size / 2 + 6, size - 1), (size / 2, size - 12)], fill=(60, 60, 60))
        _draw_on(top, fn)
    return front, top, top


def imhstack(im1, im2):
    """side by side, the second image resized to the first one's height (vis.py:724-737)"""
    h = im1.shape[0]
    if im2.shape[0] != h:
        w2 = max(1, int(round(im2.shape[1] * h / im2.shape[0])))
        im2 = np.asarray(Image.fromarray(im2).resize((w2, h), Image.BILINEAR))
    return np.concatenate((im1, im2), axis=1)


def imvstack(im1, im2):
    w = im1.shape[1]
    if im2.shape[1] != w:
        h2 = max(1, int(round(im2.shape[0] * w / im2.shape[1])))
        im2 = np.asarray(Image.fromarray(im2).resize((w, h2), Image.BILINEAR))
    return np.concatenate((im1, im2), axis=0)
