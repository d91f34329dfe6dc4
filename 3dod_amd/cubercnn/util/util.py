"""Small host utilities of the data path (reference: cubercnn/util/util.py:15-32,121)."""
import json
import os
import shutil


def file_parts(file_path):
    """('dir', 'name', '.ext')"""
    folder, tail = os.path.split(file_path)
    name, ext = os.path.splitext(tail)
    return folder, name, ext


def save_json(path, data):
    with open(path, 'w') as f:
        json.dump(data, f)


def load_json(path):
    with open(path, 'r') as f:
        return json.load(f)


def mkdir_if_missing(directory, delete_if_exist=False):
    if delete_if_exist and os.path.exists(directory):
        shutil.rmtree(directory)
    os.makedirs(directory, exist_ok=True)


_COLORS = ((106, 0, 228), (119, 11, 32), (165, 42, 42), (0, 0, 192), (197, 226, 255), (0, 60, 100), (0, 0, 142), (255, 77, 255),
           (153, 69, 1), (120, 166, 157), (0, 182, 199), (0, 226, 252), (182, 182, 255), (0, 0, 230), (220, 20, 60),
           (163, 255, 0), (0, 82, 0), (3, 95, 161), (0, 80, 100), (183, 130, 88))


def get_color(ind=None, hex=False):
    """a fixed palette cycled by index (reference: cubercnn/util/util.py get_color); random entry for ind=None"""
    import random
    c = _COLORS[ind % len(_COLORS)] if ind is not None else random.choice(_COLORS)
    return '#%02x%02x%02x' % c if hex else c


class CuboidMesh:
    """what the drawing code needs of the reference's pytorch3d `Meshes` from util.mesh_cuboid: the box, its pose, a colour"""

    def __init__(self, bbox3D, pose, color):
        self.bbox3D, self.pose, self.color = list(bbox3D), pose, color


def mesh_cuboid(box3d=None, R=None, color=None):
    """stand-in of util.mesh_cuboid (reference: cubercnn/util/math_util.py): box3d = [X,Y,Z,W,H,L], R 3x3"""
    return CuboidMesh(box3d if box3d is not None else [0, 0, 0, 1, 1, 1], R if R is not None else [[1, 0, 0], [0, 1, 0], [0, 0, 1]],
                      color)


def imread(path):
    """BGR uint8 array, like cv2.imread"""
    import numpy as np
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])


def imwrite(im, path):
    """writes a BGR uint8 array, like cv2.imwrite(path, im) with the reference's argument order (im, path)"""
    import numpy as np
    from PIL import Image
    mkdir_if_missing(os.path.dirname(os.path.abspath(path)))
    Image.fromarray(np.ascontiguousarray(np.asarray(im, dtype=np.uint8)[:, :, ::-1])).save(path)
