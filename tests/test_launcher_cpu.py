"""The N > 1 launch path on the CPU: `bench.py --gpus N` starts its ranks through
protocols.secure_comparison_amd.launcher, refuses rank counts it cannot honour, and never prints a one-rank line for an
N-rank request."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

WORKER = os.path.join(ROOT, "tests", "_launch_worker.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("B", [6, 5])  # even and ragged shards
def test_spawned_ranks_form_one_group(tmp_path, B, monkeypatch):
    from protocols.secure_comparison_amd import launcher

    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    out = tmp_path / "r.json"
    rc = launcher.spawn_ranks(WORKER, [str(out), "2", str(B)], 2, need_gpus=False)
    assert rc == 0
    got = json.load(open(out))
    assert got == {"ranks_seen": 2, "world": 2, "equal": True, "decrypts": True}


def test_failing_rank_fails_the_job(tmp_path, monkeypatch):
    from protocols.secure_comparison_amd import launcher

    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    rc = launcher.spawn_ranks(WORKER, [str(tmp_path / "r.json"), "2", "4", "1"], 2, need_gpus=False)
    assert rc != 0 and not (tmp_path / "r.json").exists()


def test_world_size_must_match_the_request(monkeypatch):
    from protocols.secure_comparison_amd import launcher

    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit):
        launcher.expect_world(8)
    assert launcher.expect_world(1) == (0, 0, 1)
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "3")
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert launcher.expect_world(4) == (3, 3, 4)
    monkeypatch.delenv("WORLD_SIZE")
    assert launcher.rank_env() is None and launcher.expect_world(1) == (0, 0, 1)


def test_bench_refuses_rank_counts_it_cannot_honour():
    """No GPU in the CPU tier: `--gpus 2` must fail before anything runs (here: this node shows fewer than 2 GPUs), and under
    a launcher whose WORLD_SIZE differs from --gpus it must fail too -- never a silent n_gpus: 1 line."""
    import torch

    bench = os.path.join(ROOT, "bench.py")
    if torch.cuda.device_count() < 2:
        cp = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_clean_env(), capture_output=True,
                            text=True, timeout=300)
        assert cp.returncode != 0 and "GPU" in cp.stderr and "n_gpus" not in cp.stdout
    env = _clean_env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    cp = subprocess.run([sys.executable, bench, "--gpus", "8", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True,
                        timeout=300)
    assert cp.returncode != 0 and "WORLD_SIZE" in cp.stderr and "n_gpus" not in cp.stdout


def test_gpu_count_comes_from_the_driver_files_not_from_hip(tmp_path, monkeypatch):
    """The spawning parent counts GPUs from /sys/class/kfd-style topology files (CPU nodes have simd_count 0) and narrows the
    count by the *_VISIBLE_DEVICES variables -- no HIP runtime is initialised for it."""
    from protocols.secure_comparison_amd import launcher

    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    assert launcher.visible_gpus(str(tmp_path)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert launcher.visible_gpus(str(tmp_path)) == 2
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert launcher.visible_gpus(str(tmp_path)) == 0
    assert launcher.host_threads_per_rank(1, 2) == min(2, os.cpu_count() or 1)
    assert launcher.host_threads_per_rank(10 ** 6, 2) == 1
    # a lease that exposes only some of the host's GPUs: sysfs still lists all of them, but only nodes whose render node
    # (/dev/dri/renderD<drm_render_minor>) this process may open count
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    dri = tmp_path / "dri"
    dri.mkdir()
    for i, minor in ((2, 128), (3, 129), (4, 130)):
        (tmp_path / str(i) / "properties").write_text(f"simd_count 1024\ndrm_render_minor {minor}\n")
    (dri / "renderD129").write_text("")
    assert launcher.visible_gpus(str(tmp_path), str(dri)) == 1
    (dri / "renderD130").write_text("")
    assert launcher.visible_gpus(str(tmp_path), str(dri)) == 2


def test_spawned_ranks_get_dmabuf_ipc_unless_the_operator_chose(tmp_path, monkeypatch):
    """HSA_ENABLE_IPC_MODE_LEGACY=0 is what RCCL needs between the ranks of a node here: the launcher sets it for its children,
    keeps a value the operator exported, and lets extra_env override both."""
    from protocols.secure_comparison_amd import launcher

    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    script = tmp_path / "w.py"
    script.write_text("import os\nopen(os.environ['OUT'] + os.environ['RANK'], 'w').write(os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', 'unset'))\n")
    out = str(tmp_path / "v")
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    assert launcher.spawn_ranks(str(script), [], 2, need_gpus=False, extra_env={"OUT": out}) == 0
    assert open(out + "0").read() == "0" and open(out + "1").read() == "0"
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")
    assert launcher.spawn_ranks(str(script), [], 1, need_gpus=False, extra_env={"OUT": out}) == 0 and open(out + "0").read() == "1"
    assert launcher.spawn_ranks(str(script), [], 1, need_gpus=False, extra_env={"OUT": out, "HSA_ENABLE_IPC_MODE_LEGACY": "0"}) == 0
    assert open(out + "0").read() == "0"


def test_stuck_rank_is_killed_and_the_failure_is_named(tmp_path, monkeypatch, capfd):
    """Rank 1 fails; rank 0 ignores SIGTERM (as a rank blocked in a collective does): the parent kills it after the grace period,
    returns rank 1's exit code and prints that rank's last stderr lines."""
    from protocols.secure_comparison_amd import launcher

    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    script = tmp_path / "w.py"
    script.write_text(
        "import os, signal, sys, time\n"
        "if os.environ['RANK'] == '1':\n"
        "    time.sleep(0.5); sys.stderr.write('rank one says: boom\\n'); sys.exit(9)\n"
        "signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"
        "time.sleep(120)\n")
    import time

    t0 = time.monotonic()
    rc = launcher.spawn_ranks(str(script), [], 2, need_gpus=False, grace_s=1.0)
    assert rc == 9 and time.monotonic() - t0 < 30
    err = capfd.readouterr().err
    assert "rank 1 exited with code 9" in err and "boom" in err


def test_job_timeout(tmp_path, monkeypatch):
    from protocols.secure_comparison_amd import launcher

    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    script = tmp_path / "w.py"
    script.write_text("import time\ntime.sleep(120)\n")
    assert launcher.spawn_ranks(str(script), [], 2, need_gpus=False, timeout_s=1.0, grace_s=1.0) == 124
