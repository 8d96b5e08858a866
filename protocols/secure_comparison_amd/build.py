"""Build libsc_amd.so (hipcc, gfx950) in-tree.  Used by __graft_entry__.build() and by hand."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "sc_lib.hip")
OUT = os.path.join(HERE, "libsc_amd.so")
DEPS = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith((".hip", ".h"))) + [
    os.path.join(HERE, "..", "..", "include", "sc_amd.h"), os.path.join(HERE, "..", "..", "include", "sc_amd_dev.h")
]


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_lib(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -pragma-unroll-threshold: the L = 27 limb-step loops (1458 multiply-adds per block) must be fully unrolled, otherwise
    # the column registers are indexed dynamically and land in scratch memory (50x slower)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-pragma-unroll-threshold=1000000",
           SRC, "-o", OUT, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stderr)
        raise subprocess.CalledProcessError(proc.returncode, cmd)
    if "loop not unrolled" in proc.stderr:
        # a partially unrolled limb loop indexes the column registers dynamically -> scratch memory -> ~50x slower
        os.remove(OUT)
        raise RuntimeError("hipcc did not fully unroll a limb loop (see -pragma-unroll-threshold in build.py)")
    return OUT


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
