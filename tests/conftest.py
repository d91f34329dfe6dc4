import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(params=["fp32", "fp32x3", "bf16"])
def precision(request):
    """runs a GPU test once per arithmetic mode of the conv path: "fp32" = the reference's precision (the default of the
    product), "fp32x3" = float32 storage with the contractions on the bf16 matrix cores through the exact three-way operand
    split (float32 accuracy: held to the fp32 tolerances), "bf16" = the opt-in fast mode.  Restores the previous mode afterwards."""
    import importlib
    ops = importlib.import_module("3dod_amd.hipops")
    prev = ops.set_precision(request.param)
    yield request.param
    ops.set_precision(prev)
