"""CPU: the C-ABI library loads and exports every symbol include/cr3dod.h declares
(no compute calls without a GPU)."""
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    return importlib.import_module("3dod_amd._lib")


def test_header_symbols_exported():
    lib_mod = _lib()
    if not os.path.exists(lib_mod.LIB_PATH):
        build = importlib.import_module("3dod_amd.build")
        build.build(verbose=False)
    lib = lib_mod.load()
    hdr = open(os.path.join(ROOT, "include", "cr3dod.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(cr_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in cr3dod.h but not exported"
    # and the python binding covers every declared entry point
    bound = set(lib_mod.SIGNATURES) | {"cr_last_error"}
    assert declared <= bound, f"unbound: {declared - bound}"
    assert lib.cr_abi_version() == 5


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    geo = importlib.import_module("3dod_amd.geometry")
    lib_mod = _lib()
    with pytest.raises(lib_mod.CrError):
        geo.cuboid_corners(torch.zeros(2, 6), torch.zeros(2, 3, 3))
