"""Host-side mirror of the reference's `cubercnn` package for the Cube R-CNN forward/loss path
(same module / class / registry names, same config keys, same state-dict keys), calling the
MI355X kernels of libcr3dod.so through 3dod_amd.hipops.  Put `3dod_amd/` on PYTHONPATH to make
`import cubercnn` resolve here."""


def _cr_bootstrap():
    """This file is executing as the TOP-LEVEL package `cubercnn` (PYTHONPATH=<repo>/3dod_amd, the reference's layout):
    load the enclosing directory as the package `3dod_amd` and become an alias of `3dod_amd.cubercnn`."""
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
    pkg._adopt_toplevel("cubercnn")


if __name__ == "cubercnn":
    _cr_bootstrap()
