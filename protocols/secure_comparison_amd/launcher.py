"""One process per GPU: start N ranks of a script on one node, or check the ranks a launcher started.

The parent of `spawn_ranks` never touches the GPU (no HIP call, no `torch.cuda.is_available()`): it counts devices with
`torch.cuda.device_count()` only, starts every rank as a fresh child process with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
in its environment (the variables `torch.distributed.run` sets), waits for all of them and hands back the first non-zero
exit code.  A script launched under `torch.distributed.run` instead finds those variables already set and must agree with the
rank count it was asked for (`expect_world`) -- a mismatch is an error, never a silent one-rank run."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Sequence


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


def visible_gpus() -> int:
    """Number of GPUs this process could use, without initialising any of them."""
    import torch

    return torch.cuda.device_count()


def spawn_ranks(script: str, argv: Sequence[str], nranks: int, *, need_gpus: bool = True, extra_env: dict[str, str] | None = None,
                poll_s: float = 0.2) -> int:
    """Run `script argv` as `nranks` child processes (rank r on local device r) and return 0 when all of them succeeded,
    otherwise the first failing rank's exit code (the remaining ranks are terminated).  With `need_gpus` the node must show
    at least `nranks` GPUs, else SystemExit(2) before anything is started."""
    if nranks < 1:
        raise SystemExit("the rank count must be at least 1")
    if need_gpus:
        have = visible_gpus()
        if have < nranks:
            raise SystemExit(f"asked for {nranks} GPU rank(s) but this node shows {have} GPU(s)")
    port = free_port()
    procs = []
    for r in range(nranks):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between the ranks of a node here
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, script, *argv], env=env))
    rc = 0
    live = set(range(nranks))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for o in live:                      # a failed rank leaves the others stuck in a collective
                    procs[o].terminate()
        if live:
            time.sleep(poll_s)
    return rc
