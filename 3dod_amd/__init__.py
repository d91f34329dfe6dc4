"""3dod_amd -- MI355X-native (gfx950) hot path of luchsonice/3dod: the Cube R-CNN
forward/loss path and the 1000-cube proposal-and-scoring geometry, behind the
reference's own module / registry names.

The directory name starts with a digit (not a Python identifier), so there are two ways in:

  * `importlib.import_module("3dod_amd.<module>")` with the repository root on sys.path, or
  * the reference's own names: put THIS directory on PYTHONPATH; `import cubercnn`, `import ProposalNetwork`
    then resolve here, exactly as tools/train_net.py:39-59 and demo/demo.py:22-27 of the reference write them
    (tests/test_dropin_route.py).

Either way every module exists ONCE: the canonical name is `3dod_amd.<...>`; the top-level names `cubercnn`,
`ProposalNetwork`, `d2lite`, `depth_anything_v2`, `hipops`, `geometry`, `synthetic` are aliases of the same module
objects, provided by a meta-path finder (install_aliases) that the packages' __init__ files install when they find
themselves imported as top-level packages.
"""
import importlib
import importlib.abc
import importlib.machinery
import sys

__version__ = "0.2.0"

ALIASED = ("cubercnn", "ProposalNetwork", "d2lite", "depth_anything_v2", "hipops", "geometry", "synthetic")


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """`cubercnn.x.y` -> the module object of `3dod_amd.cubercnn.x.y` (imported on demand); never a second copy."""

    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] not in ALIASED:
            return None
        try:
            real = importlib.import_module(__name__ + "." + name)
        except ModuleNotFoundError as e:
            if e.name == __name__ + "." + name:
                return None                      # no such module here either: let the other finders answer
            raise
        spec = importlib.machinery.ModuleSpec(name, self, is_package=hasattr(real, "__path__"))
        spec._cr_real = real
        return spec

    def create_module(self, spec):
        return spec._cr_real                     # the existing module object (its own __name__ / __spec__ are kept)

    def exec_module(self, module):
        pass


def install_aliases():
    """idempotent; puts the alias finder in front of the path finders"""
    if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _AliasFinder())


def _adopt_toplevel(name):
    """called by a sub-package's __init__ that was found through PYTHONPATH=<repo>/3dod_amd and is executing as the
    top-level package `name`: install the aliases and hand sys.modules[name] over to the canonical module"""
    install_aliases()
    real = importlib.import_module(__name__ + "." + name)
    sys.modules[name] = real
    return real
