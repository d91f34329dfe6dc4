"""Depth-Anything-V2 (metric) forward on MI355X: DINOv2 ViT encoder + DPT head with the module names and state-dict keys
of the reference's `depth/metric_depth/depth_anything_v2` package, inference only."""


def _cr_bootstrap():
    """This file is executing as the TOP-LEVEL package `depth_anything_v2` (PYTHONPATH=<repo>/3dod_amd, the reference's layout):
    load the enclosing directory as the package `3dod_amd` and become an alias of `3dod_amd.depth_anything_v2`."""
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = sys.modules.get("3dod_amd")
    if pkg is None:
        spec = importlib.util.spec_from_file_location("3dod_amd", os.path.join(root, "__init__.py"),
                                                      submodule_search_locations=[root])
        pkg = importlib.util.module_from_spec(spec)
        sys.modules["3dod_amd"] = pkg
        spec.loader.exec_module(pkg)
    pkg._adopt_toplevel("depth_anything_v2")


if __name__ == "depth_anything_v2":
    _cr_bootstrap()
else:
    from .dinov2 import DINOv2, DinoVisionTransformer  # noqa: F401
    from .dpt import DepthAnythingV2, DPTHead  # noqa: F401
