"""bench.py's N > 1 control flow on the CPU: the same main() / measure() as on the GPU box with the device side
replaced by tests/_bench_dry_run.py's stand-in (gloo, the test-only engine), so that the first real multi-GPU run can only fail
on RCCL itself: ranks started by `--gpus 2` (launcher.spawn_ranks) and ranks found in a launcher's environment, the per-step
all-gather into a persistent array, the rank check, ONE line from rank 0 whose value aggregates all ranks."""
import json
import os
import subprocess
import sys

from conftest import ROOT

DRY = os.path.join(ROOT, "tests", "_bench_dry_run.py")
ARGS = ["--batch", "3", "--l", "16", "--pbits", "1024", "--dgk", "dgk_tiny_l16", "--rbits", "50", "--fb-window", "4", "--steps", "2", "--warmup", "1",
        "--no-cpu-baseline", "--no-extras", "--no-other-configs"]


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _line(stdout: str) -> dict:
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout          # exactly one JSON line for the whole job
    return json.loads(lines[0])


def test_two_ranks_started_by_bench_itself():
    cp = subprocess.run([sys.executable, DRY, "--gpus", "2"] + ARGS, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert cp.returncode == 0, cp.stderr[-2000:]
    d = _line(cp.stdout)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["rank_devices"] == [0, 1] and d["scaling"] == "weak"
    assert d["steps"] == 2 and d["warmup"] == 1 and d["config"]["batch_per_gpu"] == 3 and d["config"]["parallelism"] == "shard2"
    # value = the comparisons of ALL ranks over the slowest rank's time
    assert abs(d["value"] - 2 * 3 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]
    assert len(d["step_ms"]) >= 3 and "cpu_baseline" not in d and "interactive_protocol" not in d
    # one diagnostic row per rank (its own steps, its step without the collective, launch / clock / probe before and after, its table)
    assert [r["rank"] for r in d["per_rank"]] == [0, 1] and [r["device"] for r in d["per_rank"]] == [0, 1]
    for r in d["per_rank"]:
        assert r["solo_step_ms"] > 0 and r["step_ms_min"] <= r["step_ms_median"] <= r["step_ms_max"] and r["fixed_base_window"] == 4
        assert r["launch_ms_before"] > 0 and r["launch_ms_after"] > 0 and r["fixed_base_table_bytes"] > 0
    slowest = max(r["solo_step_ms"] for r in d["per_rank"])
    assert abs(d["weak_scaling_efficiency"] - d["value"] / (2 * 3 / (slowest * 1e-3))) < 1e-9 * d["value"] + 1e-9


def test_table_window_falls_back_when_the_gpu_is_short_of_memory():
    """Every rank asks its GPU for its FREE memory before building the fixed-base tables: a window whose tables do not fit (with a
    quarter of the free bytes left for the batch) is replaced by the largest smaller one that does, and the line says so."""
    import bench

    full = bench.fixed_base_table_bytes(20, 400, 2048, 160)
    assert full == 8455716864                                   # what round 4's line reports for window 20 (Alice's table + the CRT halves)
    assert bench.choose_fixed_base_window(20, 288 << 30, 400, 2048, 160) == (20, None)
    w, note = bench.choose_fixed_base_window(20, 9 << 30, 400, 2048, 160)
    assert w == 16 and "fell back to window 16" in note
    assert bench.choose_fixed_base_window(24, 90 << 30, 400, 2048, 160)[0] == 20
    assert bench.choose_fixed_base_window(20, None, 400, 2048, 160) == (20, None)
    env = _clean_env()
    env["SC_DRY_FREE_BYTES"] = "30000"                           # the whole dry run on a "GPU" with 30 000 free bytes: the tiny key's window-4 tables (26 KB) do not fit
    args = [a for a in ARGS]
    cp = subprocess.run([sys.executable, DRY, "--gpus", "1"] + args, env=env, capture_output=True, text=True, timeout=600)
    assert cp.returncode == 0, cp.stderr[-2000:]
    d = _line(cp.stdout)
    assert d["config"]["fixed_base_window_requested"] == 4 and d["config"]["fixed_base_window"] < 4 and "fell back" in d["config"]["fixed_base_window_note"]
    assert "fell back" in cp.stderr


def test_two_ranks_under_a_launcher_environment():
    """What `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` does: the ranks exist already."""
    from protocols.secure_comparison_amd import launcher

    port = launcher.free_port()
    procs = []
    for r in range(2):
        env = _clean_env()
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, DRY, "--gpus", "2"] + ARGS, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs[0][1][-1500:] + outs[1][1][-1500:]
    d = _line(outs[0][0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]          # rank 1 prints nothing
    # a launcher whose WORLD_SIZE differs from --gpus is refused before any work
    env = _clean_env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    cp = subprocess.run([sys.executable, DRY, "--gpus", "2"] + ARGS, env=env, capture_output=True, text=True, timeout=300)
    assert cp.returncode != 0 and "WORLD_SIZE" in cp.stderr and "{" not in cp.stdout
