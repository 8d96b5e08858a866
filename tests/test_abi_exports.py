"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/sc_amd.h declares.
No compute call is made here (there is no GPU in the CPU test tier)."""
import ctypes
import os
import re

from conftest import ROOT
from protocols.secure_comparison_amd import _lib
from protocols.secure_comparison_amd.build import build_lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sc_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_loads_and_exports_everything():
    path = build_lib(verbose=False)
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/sc_amd.h but not exported"
    assert sorted(_lib.SYMBOLS) == names          # the ctypes binding covers the whole header
    assert _lib.load().sc_abi_version() == _lib.ABI_VERSION == 3


def test_no_gpu_means_loud_failure():
    import torch

    if torch.cuda.is_available():
        return
    lib = _lib.load()
    ctx = ctypes.c_void_p()
    assert lib.sc_ctx_create(0, ctypes.byref(ctx)) != 0      # no silent CPU fallback
    from protocols.secure_comparison_amd.engine import Engine, ScError
    import pytest

    with pytest.raises(ScError):
        Engine()
