"""One process per GPU: start N ranks of a script on one node, or check the ranks a launcher started.

The parent of `spawn_ranks` never touches the GPU: it counts devices from the kernel driver's topology files
(/sys/class/kfd), not through HIP or torch, starts every rank as a fresh child process with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* in its environment (the variables `torch.distributed.run` sets), waits for all of them and hands back the first
non-zero exit code together with that rank's last stderr lines.  A script launched under `torch.distributed.run` instead finds
those variables already set and must agree with the rank count it was asked for (`expect_world`) -- a mismatch is an error,
never a silent one-rank run."""
from __future__ import annotations

import glob
import os
import socket
import subprocess
import sys
import tempfile
import time
from typing import Sequence

RENDEZVOUS_TIMEOUT_S = 300       # ranks give up on a rendezvous / collective after this long (init_process_group timeout)
TERMINATE_GRACE_S = 10.0         # a rank that ignores SIGTERM this long (stuck in a collective) is killed


def rank_env() -> tuple[int, int, int] | None:
    """(rank, local_rank, world_size) when a launcher set them, else None."""
    if "WORLD_SIZE" not in os.environ:
        return None
    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ.get("RANK", "0"))
    return rank, int(os.environ.get("LOCAL_RANK", str(rank))), world


def expect_world(asked: int) -> tuple[int, int, int]:
    """The launcher's (rank, local_rank, world); SystemExit(2) when it differs from the rank count asked for."""
    env = rank_env()
    rank, local_rank, world = env if env is not None else (0, 0, 1)
    if world != asked:
        raise SystemExit(f"asked for {asked} rank(s) but the launcher's WORLD_SIZE is {world}: refusing to run a different job")
    if not 0 <= rank < world:
        raise SystemExit(f"RANK={rank} outside WORLD_SIZE={world}")
    return rank, local_rank, world


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _kfd_gpu_nodes(root: str = "/sys/class/kfd/kfd/topology/nodes", dri_root: str = "/dev/dri") -> int | None:
    """GPU nodes the amdgpu driver lists AND this process may open: nodes with SIMDs (CPU nodes have simd_count 0) whose render
    node /dev/dri/renderD<drm_render_minor> is readable and writable here -- sysfs lists every GPU of the host even when a
    container or lease exposes only some of their device files.  A node without a drm_render_minor line counts (older drivers).
    None when the topology files are absent."""
    files = glob.glob(os.path.join(root, "*", "properties"))
    if not files:
        return None
    n = 0
    for f in files:
        simds, minor = 0, None
        try:
            for line in open(f):
                parts = line.split()
                if len(parts) >= 2 and parts[0] == "simd_count":
                    simds = int(parts[1])
                elif len(parts) >= 2 and parts[0] == "drm_render_minor":
                    minor = int(parts[1])
        except (OSError, ValueError):
            continue
        if simds <= 0:
            continue
        if minor is not None and minor > 0 and not os.access(os.path.join(dri_root, f"renderD{minor}"), os.R_OK | os.W_OK):
            continue
        n += 1
    return n


def visible_gpus(kfd_root: str = "/sys/class/kfd/kfd/topology/nodes", dri_root: str = "/dev/dri") -> int:
    """Number of GPUs this process's children could use, WITHOUT initialising the HIP / HSA runtime in this process: the
    driver's topology files, narrowed by the *_VISIBLE_DEVICES variables.  Only when those files do not exist (no amdgpu
    driver: the CPU test tier) does it ask torch, whose answer is 0 there without loading a runtime."""
    n = _kfd_gpu_nodes(kfd_root, dri_root)
    if n is None:
        import torch

        return torch.cuda.device_count()
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def host_threads_per_rank(world: int, want: int) -> int:
    """Concurrent shard threads a rank may run: `want`, capped so that all ranks of the node together stay within its cores
    (8 ranks x (2 shard threads + the launch thread) would oversubscribe a small host)."""
    cores = os.cpu_count() or 1
    return max(1, min(want, cores // max(1, world)))


def _tail(path: str, lines: int = 15) -> str:
    try:
        with open(path, "rb") as f:
            f.seek(0, os.SEEK_END)
            size = f.tell()
            f.seek(max(0, size - 8192))
            return "\n".join(f.read().decode(errors="replace").splitlines()[-lines:])
    except OSError:
        return ""


def spawn_ranks(script: str, argv: Sequence[str], nranks: int, *, need_gpus: bool = True, extra_env: dict[str, str] | None = None,
                poll_s: float = 0.2, timeout_s: float | None = None, grace_s: float = TERMINATE_GRACE_S) -> int:
    """Run `script argv` as `nranks` child processes (rank r on local device r) and return 0 when all of them succeeded,
    otherwise the first failing rank's exit code: the remaining ranks are terminated, killed if they do not exit within
    `grace_s`, and the failing rank's last stderr lines are printed by the parent.  `timeout_s` bounds the whole job (exit code
    124).  With `need_gpus` the node must show at least `nranks` GPUs, else SystemExit(2) before anything is started.
    The children get HSA_ENABLE_IPC_MODE_LEGACY=0 unless the operator exported another value (dmabuf IPC: what RCCL needs between
    the ranks of a node on hosts whose driver supports nothing else -- without it rank-to-rank buffer sharing fails with
    `hipIpcGetMemHandle: invalid argument`); `extra_env` overrides anything."""
    if nranks < 1:
        raise SystemExit("the rank count must be at least 1")
    if need_gpus:
        have = visible_gpus()
        if have < nranks:
            raise SystemExit(f"asked for {nranks} GPU rank(s) but this node shows {have} GPU(s)")
    port = free_port()
    procs, logs = [], []
    tmp = tempfile.TemporaryDirectory(prefix="sc_ranks_")
    try:
        for r in range(nranks):
            env = dict(os.environ)
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), SC_AMD_RENDEZVOUS_TIMEOUT_S=str(RENDEZVOUS_TIMEOUT_S))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            env.setdefault("GPU_MAX_HW_QUEUES", "32")         # a hardware queue per stream of the rank (engine.py::_default_hw_queues)
            if extra_env:
                env.update(extra_env)
            log = os.path.join(tmp.name, f"rank{r}.stderr")
            logs.append(log)
            with open(log, "wb") as lf:
                procs.append(subprocess.Popen([sys.executable, script, *argv], env=env, stderr=lf))
        rc, failed = 0, -1
        live = set(range(nranks))
        t0 = time.monotonic()
        deadline_kill: float | None = None
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc, failed = (code if code > 0 else 1), r
                    for o in live:                      # a failed rank leaves the others stuck in a collective
                        procs[o].terminate()
                    deadline_kill = time.monotonic() + grace_s
            now = time.monotonic()
            if live and rc == 0 and timeout_s is not None and now - t0 > timeout_s:
                rc, failed = 124, -1
                sys.stderr.write(f"[launcher] job exceeded {timeout_s:.0f} s: terminating {len(live)} rank(s)\n")
                for o in live:
                    procs[o].terminate()
                deadline_kill = now + grace_s
            if live and deadline_kill is not None and now > deadline_kill:
                for o in live:                          # SIGTERM ignored (a rank blocked inside a collective): no more waiting
                    procs[o].kill()
                deadline_kill = None
            if live:
                time.sleep(poll_s)
        for r in range(nranks):                         # pass the ranks' stderr on (rank 0 first), then name the failure
            text = _tail(logs[r], 400 if r == 0 else 40) if rc == 0 or r != failed else ""
            if text and (r == 0 or rc != 0):
                sys.stderr.write(text + "\n")
        if rc != 0 and failed >= 0:
            sys.stderr.write(f"[launcher] rank {failed} exited with code {rc}; its last stderr lines:\n{_tail(logs[failed])}\n")
        return rc
    finally:
        tmp.cleanup()
