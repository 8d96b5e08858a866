"""Build libsc_amd.so (hipcc, gfx950) in-tree.  Used by __graft_entry__.build() and by hand.

The library is eight translation units: the host side (csrc/sc_lib.hip, no device code), the instances of the two interpreter
kernels in three parts each (csrc/sc_launch_vm.hip / sc_launch_pvm.hip with -DSC_PART=0/1/2) and the remaining kernels
(csrc/sc_launch_misc.hip).  The parts compile in parallel; an object is rebuilt only when one of the files it includes changed,
so a change of host logic or policy costs seconds and a kernel change only the parts that hold that kernel.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
OUT = os.path.join(HERE, "libsc_amd.so")
INCLUDE = os.path.join(HERE, "..", "..", "include")
# -pragma-unroll-threshold: the L = 27 limb-step loops (1458 multiply-adds per block) must be fully unrolled, otherwise
# the column registers are indexed dynamically and land in scratch memory (50x slower)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-pragma-unroll-threshold=1000000"]
# (object name, source, extra flags)
UNITS = [("sc_lib", "sc_lib.hip", [])] + \
        [(f"sc_launch_vm{p}", "sc_launch_vm.hip", [f"-DSC_PART={p}"]) for p in range(3)] + \
        [(f"sc_launch_pvm{p}", "sc_launch_pvm.hip", [f"-DSC_PART={p}"]) for p in range(3)] + \
        [("sc_launch_misc", "sc_launch_misc.hip", [])]
DEPS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) + [
    os.path.join(INCLUDE, "sc_amd.h"), os.path.join(INCLUDE, "sc_amd_dev.h")]
_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _closure(path: str, seen: set[str] | None = None) -> set[str]:
    """`path` and every file it includes with quotes, transitively."""
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for inc in _INC.findall(open(path).read()):
        _closure(os.path.join(os.path.dirname(path), inc), seen)
    return seen


def _stale(obj: str, src: str) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in _closure(src) | {os.path.abspath(__file__)})


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_lib(force: bool = False, verbose: bool = True, extra_flags: tuple[str, ...] = ()) -> str:
    if not force and not extra_flags and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for name, src, flags in UNITS:
        obj, path = os.path.join(OBJ, name + ".o"), os.path.join(CSRC, src)
        if force or extra_flags or _stale(obj, path):
            jobs.append((obj, [hipcc, *FLAGS, *flags, *extra_flags, "-c", path, "-o", obj]))

    def run(job):
        obj, cmd = job
        if verbose:
            print(" ".join(cmd), flush=True)
        proc = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        if proc.returncode == 0 and "loop not unrolled" in proc.stderr:
            # a partially unrolled limb loop indexes the column registers dynamically -> scratch memory -> ~50x slower
            os.remove(obj)
            return obj, 1, proc.stderr + "\nhipcc did not fully unroll a limb loop (see -pragma-unroll-threshold in build.py)"
        return obj, proc.returncode, proc.stderr

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1, 8))) as pool:
        results = list(pool.map(run, jobs))
    failed = [(o, err) for o, rc, err in results if rc != 0]
    if failed:
        for o, err in failed:
            sys.stderr.write(f"--- {os.path.basename(o)}\n{err}\n")
            if os.path.exists(o):
                os.remove(o)
        raise RuntimeError("hipcc failed for " + ", ".join(os.path.basename(o) for o, _ in failed))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *(os.path.join(OBJ, n + ".o") for n, _, _ in UNITS), "-o", OUT, "-ldl"]
    if verbose:
        print(" ".join(link), flush=True)
    proc = subprocess.run(link, stderr=subprocess.PIPE, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stderr)
        raise subprocess.CalledProcessError(proc.returncode, link)
    return OUT


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
