"""3dod_amd -- MI355X-native (gfx950) hot path of luchsonice/3dod: the Cube R-CNN
forward/loss path and the 1000-cube proposal-and-scoring geometry, behind the
reference's own module / registry names.

The directory name starts with a digit, so import it with
    importlib.import_module("3dod_amd")
or put `3dod_amd/` itself on PYTHONPATH to get drop-in `cubercnn` and
`ProposalNetwork` packages.
"""
__version__ = "0.1.0"
