"""GPU tests added in round 5: the segmented pair launches of 3072-bit keys (the L = 14 twins) item by item against the whole
launch and against Python's pow; concurrent single comparisons coalesced into batch launches, every message against the
uncoalesced run; the new interpreter forms of the round."""
import asyncio
import contextvars
import os
import random
import sys

import pytest
import torch

from conftest import oracle_dgk, oracle_paillier

pytestmark = pytest.mark.gpu


def test_segmented_pair_launches_of_3072_bit_keys(engine, keys):
    """Round 4 cut the long pair launches of a context that shares the chip into segments also for the L = 14 twins of 1536 / 3072-bit
    moduli (`l14_rounds`, sc_modexp_shared_sq: Alice's rho^N mod N^2 on k_pvm<8,14> in eight segments, the key holder's y^p mod p^2
    on k_pvm<4,14> in three, a table slot per item) and covered them by a decrypt property only.  Here: segmented == whole, item by
    item, and == pow(rho, N, N^2) on sampled rows, on both kernels, with batch sizes inside the branch's window (more than half a
    round of resident waves)."""
    from protocols.secure_comparison_amd import Paillier

    sk = oracle_paillier(keys, 3072)
    alice_p, bob_p = Paillier(sk.n, engine=engine), Paillier(sk.n, sk.p, sk.q, engine=engine)
    rng = random.Random(14)
    n2 = sk.n * sk.n
    base = [rng.randrange(1, sk.n) for _ in range(48)] + [1, sk.n - 1]
    engine.set_latency_mode(1)
    try:
        for scheme, count in ((alice_p, 12000), (bob_p, 24000)):       # ragged on purpose: the last wave of either launch is partly filled
            rho = engine.upload(base, scheme.mod_n.nwords).repeat(count // len(base), 1).contiguous()
            before = engine.stats()
            whole = scheme.randomizer_batch(rho)
            assert engine.stats()["segmented_pair_launches"] == before["segmented_pair_launches"]      # a context that owns the chip cuts nothing
            engine.set_chip_share(2)
            try:
                seg = scheme.randomizer_batch(rho)
            finally:
                engine.set_chip_share(1)
            after = engine.stats()
            cut = after["segmented_pair_launches"] - before["segmented_pair_launches"]
            assert cut == (2 if scheme is bob_p else 1), after                                          # Alice: one pair launch; the key holder: one per prime
            assert after["pair_segments"] - before["pair_segments"] >= 3 * cut                            # 100-ms / 33-ms rounds against a 5-ms hold: many segments
            assert torch.equal(whole, seg)
            rows = [0, 1, 7, 48, 49, len(base), rho.shape[0] - 51, rho.shape[0] - 1]
            assert engine.download(seg[torch.tensor(rows, device=seg.device)]) == [pow(base[i % len(base)], sk.n, n2) for i in rows]
    finally:
        engine.set_latency_mode(0)


@pytest.mark.parametrize("pbits, dname, l, sessions", [(1024, "dgk_1024_l16", 16, 256), (2048, "dgk_2048_l32", 32, 48)])
def test_concurrent_sessions_are_coalesced_into_batch_launches(engine, keys, pbits, dname, l, sessions):
    """The reference's primary usage -- many concurrent single comparisons on one Initiator / KeyHolder pair (SC/initiator.py:69-175,
    :86-87; test/unit/test_secure_comparison.py:804-835) -- through the session coalescer: `sessions` concurrent
    perform_secure_comparison(x_i, y_i) tasks, strict warnings, run as a handful of batch launches, and EVERY message of every
    session equals what that session sends when it runs alone through the uncoalesced one-element path with the same random stream
    (which tests/test_gpu_round4.py holds against the operator path and the oracle).  All results decrypt to x_i <= y_i."""
    sys.path.insert(0, os.path.dirname(__file__))
    from _coalesce_harness import run_sessions
    from protocols.secure_comparison_amd import DGK, Paillier

    sk, dgk = oracle_paillier(keys, pbits), oracle_dgk(keys, dname)
    bob_p = Paillier(sk.n, sk.p, sk.q, engine=engine)
    bob_d = DGK(dgk.n, dgk.g, dgk.h, dgk.u, dgk.t, dgk.p, dgk.q, dgk.v_p, dgk.v_q, engine=engine, randomizer_bits=400)
    rng = random.Random(sessions)
    pairs = [(23, 42), (42, 23), (7, 7), (-5, -5), (-9, 4)] + [(rng.randrange(1 << l), rng.randrange(1 << l)) for _ in range(sessions - 5)]
    pairs[9] = (pairs[9][0], pairs[9][0] + 1)
    engine.set_latency_mode(1)
    try:
        co_res, co_sent, stats = run_sessions(pairs, l, bob_p, bob_d, coalesce=True)
        check = range(sessions)
        un_res, un_sent, _ = run_sessions(pairs, l, bob_p, bob_d, coalesce=False)
    finally:
        engine.set_latency_mode(0)
        bob_p.shut_down()
    assert co_sent.keys() == un_sent.keys() and len(co_sent) == 4 * sessions
    for i in check:
        for step in ("step_1", "step_4b", "step_4i", "step_5"):
            k = f"{step}_session_{i + 1}"
            assert co_sent[k] == un_sent[k], k
        assert co_res[i] == un_res[i]
    dec = engine.download(bob_p.decrypt_raw_batch(engine.upload(co_res, 2 * bob_p.mod_n.nwords)))
    assert dec == [int(x <= y) for x, y in pairs]
    for side in ("alice", "bob"):
        assert stats[side]["largest"] == sessions and stats[side]["fallbacks"] == 0
    assert stats["alice"]["calls"] == 3 and stats["bob"]["calls"] == 3      # step 1 / 4 / 6+7; the randomizers ahead of time / steps 2-4b / 4j+5


def test_pair_segment_policy_follows_the_measured_hold_time(engine, keys):
    """sc_ctx_set_pair_policy: the number of segments of a long pair launch on a shared chip is round(hold / hold_ms) with the hold time
    MEASURED on the device (per-op times of the kernel instance x the program's op counts) -- no shape literal: halving hold_ms about
    doubles the segments, hold_ms = 0 and launches longer than max_rounds stay whole, and every form gives the same residues."""
    from protocols.secure_comparison_amd import Paillier

    sk = oracle_paillier(keys, 2048)
    alice_p = Paillier(sk.n, engine=engine)
    rng = random.Random(8)
    base = [rng.randrange(1, sk.n) for _ in range(64)]
    rho = engine.upload(base, alice_p.mod_n.nwords).repeat(20480 // 64, 1).contiguous()       # 1280 waves of k_pvm<4,18>: 0.6 of a round
    engine.set_latency_mode(1)
    whole = alice_p.randomizer_batch(rho)
    engine.set_chip_share(2)
    counts = {}
    try:
        for hold in (0.0, 20.0, 10.0, 5.0):
            engine.set_pair_policy(hold, 2.5)
            b = engine.stats()
            got = alice_p.randomizer_batch(rho)
            a = engine.stats()
            counts[hold] = (a["segmented_pair_launches"] - b["segmented_pair_launches"], a["pair_segments"] - b["pair_segments"])
            assert torch.equal(got, whole), hold
        engine.set_pair_policy(5.0, 0.5)                       # 0.6 rounds > max_rounds: left whole
        b = engine.stats()
        assert torch.equal(alice_p.randomizer_batch(rho), whole) and engine.stats()["segmented_pair_launches"] == b["segmented_pair_launches"]
    finally:
        engine.set_pair_policy(5.0, 2.5)
        engine.set_chip_share(1)
        engine.set_latency_mode(0)
    assert counts[0.0] == (0, 0)
    segs = {h: c[1] for h, c in counts.items() if h > 0}
    assert all(counts[h][0] == 1 for h in segs) and 2 <= segs[20.0] < segs[10.0] < segs[5.0] <= 16, counts
    assert 1.5 <= segs[10.0] / segs[20.0] <= 2.6 and 1.5 <= segs[5.0] / segs[10.0] <= 2.6, counts     # the hold time is ~50 ms: 2-3, 5, 10-11 segments
    assert engine.download(whole[:2]) == [pow(v, sk.n, sk.n * sk.n) for v in base[:2]]


@pytest.mark.parametrize("sessions,chunks", [(1, 3), (2, 1)])
def test_two_os_processes_share_the_gpu_over_a_socket(tmp_path, sessions, chunks):
    """The reference's players are separate processes (SC/test/integration/test_pool.py:41-73).  tools/gpu_two_process.py starts the
    key holder and the initiator as two fresh OS processes -- own HIP contexts on this one GPU, a Unix socket between them
    (communicator.StreamCommunicator, wire.py byte messages into pinned buffers), draws on each party's device generator -- and
    the key holder's process decrypts the initiator's results: every row x <= y.  sessions = 2: two free-running sessions per process,
    each with its own thread, library context and socket.  A small batch here; the throughput figures are in
    profiles/r05_two_process.txt."""
    import json
    import subprocess

    from conftest import ROOT

    cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_two_process.py"), "--batch", "1536", "--l", "16", "--pbits", "2048",
                         "--batches", "2", "--chunks", str(chunks), "--window", "8", "--sessions", str(sessions)], capture_output=True, text=True, timeout=600)
    assert cp.returncode == 0, cp.stderr[-2000:]
    line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["two_process"] and d["rows_checked"] == 1536 and d["rows_decrypting_to_x_le_y"] == 1536 and d["value"] > 0


def test_concurrent_single_comparisons_between_two_os_processes():
    """The reference's call shape between two OS processes on the GPU: tools/gpu_two_process_sessions.py runs 48 concurrent
    `perform_secure_comparison` sessions per burst on one Initiator and one KeyHolder in separate processes (own HIP contexts), the
    ciphertext messages as bytes over a Unix socket, each side coalescing its sessions' steps; the key holder's process decrypts the
    initiator's results: every row [x <= y], and the sessions really shared launches."""
    import json
    import subprocess

    from conftest import ROOT

    cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_two_process_sessions.py"), "--sessions", "48", "--bursts", "2", "--l", "16",
                         "--pbits", "1024"], capture_output=True, text=True, timeout=600)
    assert cp.returncode == 0, cp.stderr[-2000:]
    d = json.loads([ln for ln in cp.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["two_process_sessions"] and d["rows_checked"] == 48 and d["rows_decrypting_to_x_le_y"] == 48
    for side in ("initiator_batched_calls", "keyholder_batched_calls"):
        assert d[side]["items"] == 3 * 48 * 3 and d[side]["fallbacks"] == 0 and d[side]["calls"] < d[side]["items"] // 4, d[side]


def test_two_ranks_rehearse_the_multi_gpu_bench_on_one_gpu():
    """`SC_BENCH_SHARE_GPU=1 python bench.py --gpus 2`: two OS-process ranks on GPU 0 with gloo as the collective (RCCL refuses two ranks
    on one device, this box has one GPU).  Everything else is the N > 1 path as the driver will run it on eight GPUs -- the parent
    spawns the ranks, every rank runs its B comparisons through the real kernels, the per-step gather fills the persistent array
    inside the timed region, every rank decrypts ITS block of the gathered array, rank 0's line carries a row per rank -- and the
    line says that it is a rehearsal."""
    import json
    import subprocess

    from conftest import ROOT

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SC_BENCH_SHARE_GPU"] = "1"
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "1024", "--l", "16", "--dgk", "dgk_2048_l16", "--fb-window", "8",
                         "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=900)
    assert cp.returncode == 0, cp.stderr[-3000:]
    line = json.loads([ln for ln in cp.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["rank_devices"] == [0, 0] and "rehearsal" in line
    assert line["value"] > 0 and line["scaling"] == "weak" and line["config"]["batch_per_gpu"] == 1024
    rows = line["per_rank"]
    assert [r["rank"] for r in rows] == [0, 1] and all(r["solo_step_ms"] > 0 and r["launch_ms_before"] > 0 for r in rows)
    assert line["weak_scaling_efficiency"] > 0            # (no meaning here: the ranks' solo steps compete for the one GPU)
