"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/sc_amd.h declares.
No compute call is made here (there is no GPU in the CPU test tier)."""
import ctypes
import os
import re

from conftest import ROOT
from protocols.secure_comparison_amd import _lib
from protocols.secure_comparison_amd.build import build_lib


def declared_symbols(headers=("sc_amd.h", "sc_amd_dev.h")):
    names = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_builds_loads_and_exports_everything():
    path = build_lib(verbose=False)
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert sorted(_lib.SYMBOLS) == names          # the ctypes binding covers both headers
    assert _lib.load().sc_abi_version() == _lib.ABI_VERSION == 5
    # what a maintainer binds (sc_amd.h) holds no primitive, policy switch or probe: those are sc_amd_dev.h's
    public = declared_symbols(("sc_amd.h",))
    assert not [n for n in public if n.startswith(("sc_mod", "sc_exp_", "sc_const_", "sc_fbt_", "sc_ctx_set_latency", "sc_ctx_set_onelane",
                                                   "sc_ctx_set_chip", "sc_ctx_set_fork", "sc_ctx_set_pair", "sc_ctx_stats", "sc_peak", "sc_mac", "sc_table", "sc_clock", "sc_crt", "sc_plain"))]
    assert {"sc_initiator_step1", "sc_keyholder_step2_4b", "sc_initiator_step4", "sc_keyholder_step4j_5", "sc_initiator_step67",
            "sc_paillier_randomize", "sc_paillier_decrypt", "sc_dgk_randomize", "sc_dgk_any_zero", "sc_rng_below", "sc_allgather"} <= set(public)


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


def test_header_is_plain_c_and_links(tmp_path):
    """include/sc_amd.h is the boundary a C / cgo / JNI host binds: it must compile as strict C99 (no C++-isms outside the
    extern "C" guard) and a C program must link against libsc_amd.so and reach an entry point without Python or a GPU."""
    import shutil
    import subprocess

    from protocols.secure_comparison_amd import _lib
    from protocols.secure_comparison_amd.build import OUT

    if shutil.which("gcc") is None:
        import pytest

        pytest.skip("no gcc here")
    src = tmp_path / "host.c"
    src.write_text('#include "sc_amd.h"\n#include "sc_amd_dev.h"\n#include <stdio.h>\n'
                   'int main(void) { printf("%d\\n", sc_abi_version()); return 0; }\n')
    exe = tmp_path / "host"
    libdir = os.path.dirname(OUT)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-lsc_amd", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert int(out.strip()) == _lib.ABI_VERSION
