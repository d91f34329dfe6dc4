"""Host-side mirror of the reference's `ProposalNetwork` package (the 1000-cube proposal-and-scoring method)
over the fused geometry kernels of libcr3dod.so.  Same module / function names as the reference:
ProposalNetwork.utils.spaces.Cubes, .utils.conversions.cubes_to_box, .proposals.proposals.propose,
.scoring.scorefunction.score_*, .utils.plane.Plane."""


def _cr_bootstrap():
    """This file is executing as the TOP-LEVEL package `ProposalNetwork` (PYTHONPATH=<repo>/3dod_amd, the reference's layout):
    load the enclosing directory as the package `3dod_amd` and become an alias of `3dod_amd.ProposalNetwork`."""
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
    pkg._adopt_toplevel("ProposalNetwork")


if __name__ == "ProposalNetwork":
    _cr_bootstrap()
