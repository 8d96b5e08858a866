"""bench.py -- secure comparisons / s on MI355X (BASELINE.json metric), one process per GPU.

A "step" is one full batch of B independent secure comparisons (both parties' compute, every randomization and the
step-4i shuffle) on synthetic inputs already resident in HBM.  Default workload = BASELINE.json configs[2]:
B = 65536, l = 32, 2048-bit Paillier + 2048-bit DGK on one GPU.  With N > 1 ranks each rank runs its own shard of B
comparisons (weak scaling, no data-path collective) and the [[x<=y]] results are all-gathered over RCCL.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--l 32] [--pbits 2048]

`--gpus N` with N > 1 and no launcher environment starts N rank processes itself (protocols.secure_comparison_amd.launcher:
the parent never touches the GPU); under `python -m torch.distributed.run --nproc-per-node N` the launcher's WORLD_SIZE must
equal N.  Either way a rank count that cannot be honoured is an error, not a one-rank run.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import subprocess
import sys
import tempfile
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")   # before the HIP runtime initialises: see protocols/secure_comparison_amd/engine.py::_default_hw_queues

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KEYS = os.path.join(ROOT, "tests", "golden", "keys.json")
MAX_CLOCK_HZ = 2.4e9   # MI355X peak engine clock (MI355X_MICROARCH.md)
DEFAULT_FB_WINDOW = 20  # fixed-base window of the tables for h: 6.2 GB per GPU of 288 (21 table products per DGK randomizer instead of the 26 of
                        # window 16 / 0.66 GB: +1.0 % on the whole step in a same-box A/B, profiles/r04_ab_vs_round2_tag.txt; window_sensitivity in the line).
                        # Window 24 (17 products from an 82 GB table, built in 1.6 .. 3.6 s) is supported and was measured: +0.35 % in an alternated
                        # A/B -- less than the four saved products of 71 would give; presumably row fetches that miss the address translation caches -- and the configs[1] sub-line beside the
                        # 82 GB drops from 114 k to 94 k/s (profiles/r04_fixed_base_window_24.txt): not the default
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def literal_macs_per_comparison(l: int, pbits: int, dbits: int, rbits: int) -> float:
    """SURVEY 8(d) literal op mix in 32x32->64 multiply-accumulates (window-5 modexp, comb-8 fixed base)."""
    def mm(bits):  # CIOS Montgomery product on s 32-bit limbs
        s = bits // 32
        return 2 * s * s + s
    def e_var(kbits, modbits):
        return (kbits + -(-kbits // 5) + 30) * mm(modbits)
    e_p = e_var(pbits, 2 * pbits)
    e_h = -(-rbits // 8) * mm(dbits)
    e_rho = (l + 3 + -(-(l + 3) // 5) + 10) * mm(dbits)   # (l+3)-bit blinding exponent, about 52 modmuls at l = 32
    e_0 = e_var(160, dbits // 2)
    m_n = mm(dbits)
    return 5 * e_p + 2 * (l + 1) * e_h + (l + 1) * e_rho + (l + 1) * e_0 + l * (l - 1) / 2 * m_n + 9 * l * m_n + 3 * 2.5 * l * m_n


def algorithmic_bytes_per_comparison(l: int, pbits: int, dbits: int, rbits: int) -> float:
    """Unavoidable HBM I/O of the step inputs/outputs: both parties' per-bit DGK vectors in and out,
    nine Paillier ciphertexts, and the randomness."""
    ct_d, ct_p = dbits // 8, 2 * pbits // 8
    rand = 2 * (l + 1) * (rbits // 8) + (l + 1) * 8 + 5 * (pbits // 8)
    return 2 * (2 * (l + 1) * ct_d) + 9 * ct_p + rand


def fixed_base_table_bytes(window: int, rbits: int, dgk_bits: int, t_bits: int, keyholder_crt: bool = True) -> int:
    """Device bytes of the fixed-base tables one GPU holds for a DGK key at `window`: Alice's rows for h modulo n (ceil(rbits / w)
    windows of 2^w rows) plus, for a key holder that randomizes through CRT, the half-size rows for h modulo p and q (t-bit exponents,
    window capped at 20: sc_dgk_key_create).  Rows are limb form: 29-bit limbs in 32-bit words, the configuration's S = G L limbs."""
    def row_bytes(bits):
        for cap, limbs in ((522, 18), (1044, 36), (1566, 54), (2088, 72), (3132, 108), (4176, 144), (6264, 216), (8352, 288)):
            if bits + 8 <= cap:
                return 4 * limbs
        raise ValueError(f"no configuration for a {bits}-bit modulus")

    nwin = lambda bits, w: -(-bits // w)  # noqa: E731
    total = nwin(rbits, window) * (1 << window) * row_bytes(dgk_bits)
    if keyholder_crt:
        w2 = min(window, 20)
        total += 2 * nwin(t_bits, w2) * (1 << w2) * row_bytes(dgk_bits // 2)
    else:
        total *= 2                      # a key holder without CRT builds the same table for h modulo n as Alice
    return total


def choose_fixed_base_window(requested: int, free_bytes: int | None, rbits: int, dgk_bits: int, t_bits: int, keyholder_crt: bool = True,
                             keep_free_fraction: float = 0.25):
    """(window, note): the requested window when its tables fit the GPU's FREE memory with room to spare for the batch itself (a
    quarter of the free bytes stays untouched), else the largest smaller window that does -- never silently: the note goes into
    the line.  free_bytes None (no way to ask): the requested window."""
    if free_bytes is None:
        return requested, None
    budget = free_bytes * (1.0 - keep_free_fraction)
    need = fixed_base_table_bytes(requested, rbits, dgk_bits, t_bits, keyholder_crt)
    if need <= budget:
        return requested, None
    for w in range(requested - 1, 0, -1):
        if w > 16 and w != 20:
            continue                    # the measured points: 20, 16, then whatever fits
        got = fixed_base_table_bytes(w, rbits, dgk_bits, t_bits, keyholder_crt)
        if got <= budget:
            return w, (f"fixed-base window {requested} needs {need / 2**30:.1f} GiB of tables, the GPU has {free_bytes / 2**30:.1f} GiB free "
                       f"(a quarter of that stays free for the batch): fell back to window {w} ({got / 2**30:.2f} GiB)")
    raise SystemExit(f"bench.py: not even a window-1 fixed-base table fits the {free_bytes} free bytes of this GPU")


def synth_inputs(eng, l, alice_p, bob_p, bob_d, B, rbits, seed, shuffle=False):
    """Seeded synthetic batch, generated on the device (SURVEY 8(d) 'Synthetic inputs').  shuffle: also draw the step-4i
    permutation of every comparison (SC/initiator.py:516, do_shuffle=True)."""
    import torch

    from protocols.secure_comparison_amd.batch import BatchDraws

    g = torch.Generator(device=eng.device)
    g.manual_seed(0xC0FFEE + seed)
    nw = alice_p.mod_n.nwords
    dev = eng.device

    def rand_words(*shape, top_bits_clear=0):
        w = torch.randint(-(2 ** 31), 2 ** 31, shape, generator=g, device=dev, dtype=torch.int64).to(torch.int32)
        if top_bits_clear:
            w[..., -1] &= (1 << (32 - top_bits_clear)) - 1
        return w

    def below_n(count):  # uniform below 2^(bits-1) <= N, never zero
        w = rand_words(count, nw, top_bits_clear=1 + (32 * nw - alice_p.public_key.n.bit_length()))
        w[:, 0] |= 1
        return w

    xb = min(l, 62)  # int64 tensors: for l > 62 the synthetic inputs use the low 62 bits of the range
    x = torch.randint(0, 2 ** xb, (B,), generator=g, device=dev, dtype=torch.int64)
    y = torch.randint(0, 2 ** xb, (B,), generator=g, device=dev, dtype=torch.int64)
    sel = torch.arange(B, device=dev) % 8
    y = torch.where(sel == 0, x, y)
    y = torch.where(sel == 1, torch.clamp(x + 1, max=2 ** xb - 1), y)

    def small_words(v):
        w = torch.zeros((v.shape[0], nw), dtype=torch.int32, device=dev)
        w[:, 0] = (v & 0xFFFFFFFF).to(torch.int32)
        w[:, 1] = (v >> 32).to(torch.int32)
        return w

    x_enc = bob_p.randomize_batch(bob_p.encrypt_raw_batch(small_words(x)), below_n(B))
    y_enc = bob_p.randomize_batch(bob_p.encrypt_raw_batch(small_words(y)), below_n(B))
    u = bob_d.public_key.u
    ew, er = (u.bit_length() + 31) // 32, (rbits + 31) // 32
    rho_small = torch.randint(0, 2 ** 62, ((l + 1), B), generator=g, device=dev, dtype=torch.int64) % (u - 1) + 1 if u < 2 ** 62 else None
    rhos = torch.zeros((l + 1, B, ew), dtype=torch.int32, device=dev)
    if rho_small is not None:
        rhos[..., 0] = (rho_small & 0xFFFFFFFF).to(torch.int32)
        if ew > 1:
            rhos[..., 1] = (rho_small >> 32).to(torch.int32)
    else:  # l = 64: u has 67 bits; draw 66 random bits (always < u) and force non-zero
        rhos = rand_words(l + 1, B, ew, top_bits_clear=32 * ew - (u.bit_length() - 1))
        rhos[..., 0] |= 1
    rho_bob = below_n(3 * B)    # the key holder's three randomizer inputs as row blocks of one array (joined again without a copy)
    draws = BatchDraws(
        r=below_n(B), delta_a=torch.randint(0, 2, (B,), generator=g, device=dev, dtype=torch.int64), rhos=rhos,
        permutation=None, rho_z=below_n(B),
        r_bob_dgk=rand_words(l + 1, B, er, top_bits_clear=32 * er - rbits),
        r_alice_dgk=rand_words(l + 1, B, er, top_bits_clear=32 * er - rbits),
        rho_zeta_1=rho_bob[:B], rho_zeta_2=rho_bob[B:2 * B], rho_delta_b=rho_bob[2 * B:])
    if shuffle:   # drawn last so that the other inputs of a seed do not depend on the flag
        draws.permutation = torch.argsort(torch.rand((B, l + 1), generator=g, device=dev, dtype=torch.float64), dim=1)
    return x, y, x_enc, y_enc, draws


def export_sample(path, eng, idx, l, x_enc, y_enc, draws, result):
    """Rows `idx` of the resident batch (inputs, every random draw, and the GPU's results) as hex integers for the CPU
    baseline leg: the oracle then runs THE SAME comparisons and its results are compared with the GPU's bit for bit."""
    import torch

    it = torch.tensor(idx, device=eng.device)
    rows = lambda t: [hex(v) for v in eng.download(t[it])]                                   # noqa: E731
    planes = lambda t: [[hex(v) for v in eng.download(t[:, i])] for i in idx]                # noqa: E731
    perm = None if draws.permutation is None else draws.permutation[it].tolist()
    r_alice = planes(draws.r_alice_dgk)
    if perm is not None:
        # the library randomizes c_j with r_alice[j] before the shuffle; the oracle randomizes output k = c_{perm[k]} after it
        r_alice = [[row[src] for src in pm] for row, pm in zip(r_alice, perm)]
    M = (1 << 64) - 1
    doc = {"l": l, "x_enc": rows(x_enc), "y_enc": rows(y_enc), "r": rows(draws.r),
           "delta_a": [int(v) & M for v in draws.delta_a[it].tolist()], "rhos": planes(draws.rhos), "perm": perm,
           "rho_z": rows(draws.rho_z), "r_bob": planes(draws.r_bob_dgk), "r_alice": r_alice, "rho_zeta_1": rows(draws.rho_zeta_1),
           "rho_zeta_2": rows(draws.rho_zeta_2), "rho_delta_b": rows(draws.rho_delta_b), "gpu_result": rows(result)}
    with open(path, "w") as f:
        json.dump(doc, f)


def policy_of(eng) -> dict:
    """The measured constants behind the library's automatic policies (a calibration the library itself rejected as implausible --
    the chip was busy -- is reported as such, not raised)."""
    try:
        out = dict(eng.policy())
    except Exception as exc:  # noqa: BLE001
        out = {"error": str(exc)[:200]}
    if hasattr(eng, "stats"):
        out["context_stats"] = eng.stats()
    return out


def cpu_interpreter() -> str:
    """The interpreter the CPU oracle runs under: the one that has gmpy2 (the reference's optional fast path) when there is one."""
    return "/opt/conda/bin/python3.9" if os.path.exists("/opt/conda/bin/python3.9") else sys.executable


def run_cpu_oracle(interp: str, count: int, procs: int, pname: str, dgk_name: str, rbits: int, sample_path: str | None = None) -> dict:
    """oracle/cpu_baseline.py in a child process (the CHECKER and the reported CPU baseline; never the thing measured as `value`)."""
    cp = subprocess.run([interp, os.path.join(ROOT, "oracle", "cpu_baseline.py"), KEYS, pname, dgk_name, str(count), str(procs), str(rbits)] +
                        ([sample_path] if sample_path else []), capture_output=True, text=True, timeout=900)
    if cp.returncode != 0:
        raise RuntimeError(cp.stderr.strip().splitlines()[-1] if cp.stderr.strip() else "cpu baseline failed")
    return json.loads(cp.stdout.strip().splitlines()[-1])


def oracle_rows_check(eng, rows: int, B: int, l: int, x_enc, y_enc, draws, res, pname: str, dgk_name: str, rbits: int) -> dict:
    """`rows` evenly spaced rows of the resident batch (inputs, every draw, the GPU's results) re-run by the oracle: how many of its
    results equal the GPU's bit for bit."""
    cores = min(os.cpu_count() or 1, 16)
    stride = max(1, B // rows)
    idx = [i * stride for i in range(min(rows, B))]
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "sample.json")
        export_sample(path, eng, idx, l, x_enc, y_enc, draws, res)
        cb = run_cpu_oracle(cpu_interpreter(), len(idx), cores, pname, dgk_name, rbits, path)
    return {"rows": cb["count"], "equal_to_oracle": cb["match_gpu"], "arith": cb["arith"], "cpu_value": cb["value"], "cores": cb["cores"]}


def interactive_protocol_leg(torch, eng, parties, ns, shard_inputs, x_enc, y_enc, expect, l, B, headline, more_parties=None):
    """Throughput of the batched INTERACTIVE protocol with draws=None and where its time goes.  `sessions` concurrent sessions
    (one per shard context, each its own thread / stream / event loop -- the reference's session_id-namespaced parallel runs,
    SC/test/unit/test_secure_comparison.py:803-835) mirror the headline's concurrent shards."""
    import asyncio
    from concurrent.futures import ThreadPoolExecutor

    from protocols.secure_comparison_amd import Initiator, KeyHolder, wire
    from protocols.secure_comparison_amd.batch import draw_alice, draw_bob
    from protocols.secure_comparison_amd.communicator import InMemoryCommunicator

    def run(device_tensors: bool, sessions: int, reps: int, chunks: int = 1):
        inputs = [(x_enc, y_enc)] if sessions == 1 else [(si[0], si[1]) for si in shard_inputs]
        players = []
        for ps in parties[:sessions]:
            comm = InMemoryCommunicator(device_tensors=device_tensors)
            # both players' scheme objects exist already (tables built: untimed set-up, as for the headline)
            players.append((Initiator(l, comm, "keyholder", ps.alice_paillier, ps.alice_dgk),
                            KeyHolder(l, comm.peer(), "initiator", ps.bob_paillier, ps.bob_dgk), ps))
        caller = torch.cuda.current_stream()
        for _, _, ps in players:      # concurrent sessions share the chip (batch-size policies of the library), like concurrent shards
            ps.alice_paillier.engine.set_chip_share(sessions)

        def session(i):
            alice, bob, ps = players[i]

            async def go():
                res, _ = await asyncio.gather(alice.perform_secure_comparison_batch(*inputs[i], engine=ps.alice_paillier.engine, chunks=chunks),
                                              bob.perform_secure_comparison_batch())
                return res

            if sessions == 1:
                return asyncio.run(go())
            with torch.cuda.device(eng.device), torch.cuda.stream(ps.stream):
                ps.stream.wait_stream(caller)
                res = asyncio.run(go())
                res.record_stream(caller)
                return res

        def once():
            if sessions == 1:
                return session(0)
            with ThreadPoolExecutor(max_workers=sessions) as pool:
                parts = list(pool.map(session, range(sessions)))
            for _, _, ps in players:
                caller.wait_stream(ps.stream)
            return torch.cat(parts, dim=0)

        res = once()                                   # warm-up (program caches of the generator-sized launches)
        torch.cuda.synchronize()
        wire.reset_stats()
        t0 = time.perf_counter()
        for _ in range(reps):
            res = once()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        for _, _, ps in players:
            ps.alice_paillier.engine.set_chip_share(1)
        dec = parties[0].bob_paillier.decrypt_raw_batch(res.contiguous())
        ok = bool(((dec[:, 0] == expect) & (dec[:, 1:] == 0).all(dim=1)).all().item())
        return dt, ok, {k: (v / reps) for k, v in wire.STATS.items()}

    # the generator alone: all draws of one batch, both parties, timed with HIP events on the library's stream
    ps0 = parties[0]
    draw_alice(B, l, ps0.alice_paillier, ps0.alice_dgk), draw_bob(B, l, ps0.bob_paillier, ps0.bob_dgk)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    da, db = draw_alice(B, l, ps0.alice_paillier, ps0.alice_dgk), draw_bob(B, l, ps0.bob_paillier, ps0.bob_dgk)
    e1.record()
    torch.cuda.synchronize()
    rng_s = e0.elapsed_time(e1) * 1e-3
    rng_bytes = sum(t.numel() * t.element_size() for d in (da, db) for t in vars(d).values() if t is not None)
    del da, db
    dt1, ok1, _ = run(True, 1, 2)
    dtn, okn, _ = (run(True, ns, 2) if ns > 1 else (dt1, ok1, None))
    dth, okh, st = run(False, 1, 1)

    def stream_of_batches(sessions: int, chunks: int, reps: int):
        """The byte transport as a deployment runs it: `sessions` sessions on one connection pair each, every session sending batch
        after batch of B / sessions comparisons without a barrier between them, started a fraction of a batch apart so that one
        session's messages drain over PCIe while the others compute (sessions started together stay in lock step: they reach
        their transfer phases at the same time).  Throughput of the steady window: the session-batches completed after every
        session has finished its first one, over the time from that moment to the last completion."""
        import threading

        sets = list(parties)
        while len(sets) < sessions and more_parties is not None:
            sets.append(more_parties())
        if len(sets) < sessions:
            return None
        per = B // sessions
        inputs = [(x_enc[i * per:(i + 1) * per].contiguous(), y_enc[i * per:(i + 1) * per].contiguous()) for i in range(sessions)]
        for ps in sets[:sessions]:
            ps.alice_paillier.engine.set_chip_share(sessions)
        torch.cuda.synchronize()
        batch_s = B / headline if headline else 0.4            # what one round of `sessions` session-batches takes on this GPU
        done, last, errors = [], [None] * sessions, []
        t0 = time.perf_counter()

        def session(i):
            ps = sets[i]
            comm = InMemoryCommunicator(device_tensors=False)
            alice = Initiator(l, comm, "keyholder", ps.alice_paillier, ps.alice_dgk)
            bob = KeyHolder(l, comm.peer(), "initiator", ps.bob_paillier, ps.bob_dgk)
            try:
                time.sleep(i * batch_s / sessions)
                with torch.cuda.device(eng.device), torch.cuda.stream(ps.stream):
                    for _ in range(reps):
                        async def go():
                            res, _ = await asyncio.gather(alice.perform_secure_comparison_batch(*inputs[i], engine=ps.alice_paillier.engine, chunks=chunks),
                                                          bob.perform_secure_comparison_batch())
                            return res

                        last[i] = asyncio.run(go())
                        ps.stream.synchronize()
                        done.append(time.perf_counter() - t0)
            except Exception as exc:  # pragma: no cover
                errors.append(repr(exc))

        threads = [threading.Thread(target=session, args=(i,)) for i in range(sessions)]
        [t.start() for t in threads]
        [t.join() for t in threads]
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        for ps in sets[:sessions]:
            ps.alice_paillier.engine.set_chip_share(1)
        if errors or any(r is None for r in last):
            return {"error": (errors or ["a session returned nothing"])[0][:200]}
        dec = parties[0].bob_paillier.decrypt_raw_batch(torch.cat(last, dim=0))
        ok = bool(((dec[:, 0] == expect[:per * sessions]) & (dec[:, 1:] == 0).all(dim=1)).all().item())
        done.sort()
        n, span = len(done) - sessions, done[-1] - done[sessions - 1]
        steady = n * per / span if n > 0 and span > 0 else sessions * reps * per / total
        return {"value": steady, "ratio_to_headline": steady / (headline if headline else 1.0), "correct": ok, "sessions": sessions, "chunks": chunks,
                "batches_per_session": reps, "comparisons_per_session_batch": per, "whole_run_value": sessions * reps * per / total,
                "steady_window_ms": span * 1e3, "session_batches_in_window": n}

    # the same byte transport pipelined: free-running concurrent sessions (one session's messages drain over PCIe while the others
    # compute) and, inside each session, optionally the batch cut into chunks whose messages are packed on a copy stream
    piped = {}
    stream_of_batches(ns, 1, 1)                             # staging buffers, program caches
    # the number of record: `ns` free-running sessions for TWENTY batches each (round 4 quoted the steady window of a four-batch run: six
    # session-batches in one second); the variants stay short
    for name, (sess, chunks, reps) in (("sessions_%d" % ns, (ns, 1, 20)), ("sessions_%d" % (ns + 1), (ns + 1, 1, 4)), ("sessions_%d_chunks_2" % ns, (ns, 2, 4))):
        r_ = stream_of_batches(sess, chunks, reps)
        if r_ is not None:
            piped[name] = r_
    record = piped.get("sessions_%d" % ns)
    best = record if record is not None and "value" in record else None
    return {
        "value": B / dtn, "unit": "comparisons/s", "ratio_to_headline": B / dtn / (headline if headline else 1.0),
        "sessions": ns, "correct": ok1 and okn and okh and all(v.get("correct", False) for v in piped.values()),
        "single_session": {"value": B / dt1, "ms_per_batch": dt1 * 1e3},
        "split_ms_per_batch": {"device_rng": rng_s * 1e3, "wire_pack_unpack": 0.0, "everything_else_gpu_and_host": (dt1 - rng_s) * 1e3},
        "device_rng": {"bytes_per_comparison": rng_bytes / B, "GB_per_s": rng_bytes / rng_s / 1e9,
                       "generator": "ChaCha20 block function (RFC 8439) in counter mode, keyed per context from the OS; rejection sampling, coins and shuffles on the device"},
        "byte_transport": {"value": best["whole_run_value"] if best else B / dth,
                           "ratio_to_headline": (best["whole_run_value"] if best else B / dth) / (headline if headline else 1.0),
                           "steady_window_value": best["value"] if best else None,
                           "pipelined": piped,
                           "single_session_unpipelined": {"value": B / dth, "ms_per_batch": dth * 1e3},
                           "ms_per_batch": dth * 1e3, "wire_pack_ms": st["pack_s"] * 1e3, "wire_unpack_ms": st["unpack_s"] * 1e3,
                           "wire_bytes_per_comparison": st["bytes"] / B,
                           "note": "same protocol with every message serialized into one pinned host buffer (one device-to-host copy per array) and "
                                   "parsed back (one host-to-device copy per array): what a transport between two processes adds.  value = the WHOLE RUN "
                                   "of %d free-running concurrent sessions (started a fraction of a batch apart) over 20 batches each, first batch and "
                                   "ramp included; steady_window_value = the same run counted from the moment every session has finished its first batch; "
                                   "ms_per_batch and the pack / unpack split are those of ONE unpipelined session; two OS processes on one GPU over a "
                                   "socket: tools/gpu_two_process.py, profiles/r05_two_process.txt" % ns},
        "note": "draws=None: all random inputs generated on the device inside the timed region; messages are the device arrays themselves "
                "(InMemoryCommunicator.device_tensors); informational, never `value`"}


def latency_single_leg(torch, eng, keys):
    """BASELINE configs[0]: ONE comparison, x = 23, y = 42, l = 16, 1024-bit Paillier + 1024-bit DGK -- the reference's primary use
    (SC/initiator.py:69-175, README.md:99-141).  Wall-clock milliseconds of the two players' perform_secure_comparison coroutines
    over the in-memory transport (scheme hand-over, the 4 + 2(l+1) randomizers generated on the spot, all four exchanges), through
    the step-level library calls on one-element batches (the default) and through the reference-shaped body that launches one
    kernel per ciphertext operator; and of the static step chain without randomization.  There is no CPU path in the product:
    this is what a batch of one costs on a GPU."""
    import asyncio
    import statistics

    from protocols.secure_comparison_amd import DGK, Initiator, KeyHolder, Paillier
    from protocols.secure_comparison_amd.batch import BatchDraws, secure_comparison_batch
    from protocols.secure_comparison_amd.communicator import InMemoryCommunicator

    pj, dj = keys["paillier_1024"], keys["dgk_1024_l16"]
    p, q, l = int(pj["p"], 16), int(pj["q"], 16), 16
    H = lambda name: int(dj[name], 16)  # noqa: E731
    bob_p = Paillier(p * q, p, q, engine=eng)
    bob_d = DGK(H("p") * H("q"), H("g"), H("h"), H("u"), dj["t"], H("p"), H("q"), H("v_p"), H("v_q"), engine=eng, randomizer_bits=400)

    def once(fused):
        comm = InMemoryCommunicator()
        alice, bob = Initiator(l, comm, "keyholder"), KeyHolder(l, comm.peer(), "initiator", bob_p, bob_d)
        # True: the default path (steps as batch launches through the session coalescer, here a batch of one); "alone": the same five
        # step-level calls without the coalescer (round 4's default); False: one launch per ciphertext operator
        alice.fuse_steps = bob.fuse_steps = bool(fused)
        alice.coalesce_sessions = bob.coalesce_sessions = fused is True

        async def go():
            res, _ = await asyncio.gather(alice.perform_secure_comparison(23, 42), bob.perform_secure_comparison())
            return res

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = asyncio.run(go())
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3, bob_p.decrypt(res) == 1

    def median_ms(fused, reps: int):
        once(fused)                                   # programs, tables, allocator
        runs = [once(fused) for _ in range(reps)]
        return statistics.median(t for t, _ in runs), all(ok for _, ok in runs)

    fused_ms, ok_f = median_ms(True, 7)
    alone_ms, ok_a = median_ms("alone", 7)
    oper_ms, ok_o = median_ms(False, 3)
    # the static step chain (README.md:117-141; no randomization) as one five-call batch of one
    alice_p, alice_d = bob_p.public_copy(), bob_d.public_copy()
    up = eng.upload
    nw, ew = alice_p.mod_n.nwords, (H("u").bit_length() + 31) // 32
    one = lambda v, w: up([v], w)  # noqa: E731
    draws = BatchDraws(r=one(12345678901234567890 % (p * q), nw), delta_a=eng.upload_u64([1]), rhos=up([3 + i for i in range(l + 1)], ew).reshape(l + 1, 1, ew),
                       permutation=None, rho_z=None, r_bob_dgk=None, r_alice_dgk=None, rho_zeta_1=None, rho_zeta_2=None, rho_delta_b=None)
    x_enc, y_enc = bob_p.encrypt_raw_batch(one(23, nw)), bob_p.encrypt_raw_batch(one(42, nw))

    def chain():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r_ = secure_comparison_batch(x_enc, y_enc, l, alice_p, alice_d, bob_p, bob_d, draws, randomize=False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3, r_

    chain()
    chain_runs = [chain() for _ in range(7)]
    dec = eng.download(bob_p.decrypt_raw_batch(chain_runs[-1][1]))[0]
    return {"workload": "BASELINE configs[0]: x=23, y=42, l=16, 1024-bit Paillier + 1024-bit DGK, one comparison",
            "interactive_ms": fused_ms, "interactive_uncoalesced_ms": alone_ms, "interactive_operator_path_ms": oper_ms,
            "static_step_chain_ms": statistics.median(t for t, _ in chain_runs),
            "correct": bool(ok_f and ok_a and ok_o and dec == 1),
            "note": "interactive_ms: both players' perform_secure_comparison over the in-memory transport through the default path -- the session coalescer "
                    "with one session in flight, i.e. five library calls on one-element batches with the randomizations fused into them; interactive_uncoalesced_ms: "
                    "round 4's default, the same five calls with the randomizers popped from the schemes' pools; every randomizer generated inside the call -- the key holder's three Paillier randomizers in the background on a second context, "
                    "like the reference's background workers (KeyHolder.background_randomness; 11.9 ms without) -- (median of 7 after a warm-up run); interactive_operator_path_ms: the same "
                    "exchange with one launch per ciphertext operator (Initiator.fuse_steps = False), identical ciphertexts; static_step_chain_ms: "
                    "steps 1-7 without randomization as a batch of one.  A single comparison is a chain of dependent launches on an otherwise idle "
                    "chip: the product has no CPU path, cpu_oracle_ms (same box, one core) is the reference point"}


def concurrent_sessions_leg(torch, eng, keys, sessions: int = 1024, shapes=(("paillier_1024", "dgk_1024_l16", 16), ("paillier_2048", "dgk_2048_l32", 32)),
                            cpu_run=None, window: int = 16):
    """The reference's primary usage at scale (SC/initiator.py:69-175, :86-87; test/unit/test_secure_comparison.py:804-835): `sessions`
    concurrent perform_secure_comparison(x_i, y_i) calls on ONE Initiator / KeyHolder pair over the in-memory transport, every one a
    full interactive comparison with its own draws and 4 + 2(l+1) randomizers.  The players' session coalescer (coalesce.py) runs the
    sessions' steps as batch launches; `uncoalesced` is the same call shape with one library call per session and step (round 4's
    path: N sessions cost N single-comparison chains).  Wall clock of the whole asyncio run, Python included."""
    import asyncio

    from protocols.secure_comparison_amd import DGK, Initiator, KeyHolder, Paillier
    from protocols.secure_comparison_amd.communicator import InMemoryCommunicator

    out = []
    for pname, dname, l in shapes:
        pj, dj = keys[pname], keys[dname]
        p, q = int(pj["p"], 16), int(pj["q"], 16)
        H = lambda name: int(dj[name], 16)  # noqa: E731
        bob_p = Paillier(p * q, p, q, engine=eng)
        bob_d = DGK(H("p") * H("q"), H("g"), H("h"), H("u"), dj["t"], H("p"), H("q"), H("v_p"), H("v_q"), engine=eng, randomizer_bits=400,
                    fixed_base_window=window)
        alice_d = bob_d.public_copy()
        alice_d.prepare(), bob_d.prepare()                 # tables: untimed set-up, like key generation
        rng = random.Random(l)
        pairs = [(rng.randrange(1 << l), rng.randrange(1 << l)) for _ in range(sessions)]
        for i in range(0, sessions, 8):
            pairs[i] = (pairs[i][0], pairs[i][0])           # equal inputs in every eighth session

        def run(n: int, coalesce: bool, pause_collector: bool = True):
            comm = InMemoryCommunicator()
            alice, bob = Initiator(l, comm, "keyholder", bob_p.public_copy(), alice_d), KeyHolder(l, comm.peer(), "initiator", bob_p, bob_d)
            alice.coalesce_sessions = bob.coalesce_sessions = coalesce
            if not pause_collector:
                alice.coalesce_pause_collector_s = bob.coalesce_pause_collector_s = 0

            async def go():
                a = [asyncio.ensure_future(alice.perform_secure_comparison(x, y)) for x, y in pairs[:n]]
                b = [asyncio.ensure_future(bob.perform_secure_comparison()) for _ in range(n)]
                res = await asyncio.gather(*a)
                await asyncio.gather(*b)
                return res

            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = asyncio.run(go())
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            dec = eng.download(bob_p.decrypt_raw_batch(eng.upload([r.peek_value() for r in res], 2 * bob_p.mod_n.nwords)))
            return dt, dec == [int(x <= y) for x, y in pairs[:n]], (alice._coalescer().stats, bob._coalescer().stats)

        run(min(64, sessions), True)                       # programs, key objects, allocator, the background context
        runs = [run(sessions, True) for _ in range(3)]
        dt = sorted(r[0] for r in runs)[1]
        from protocols.secure_comparison_amd.coalesce import quiet_collector

        with quiet_collector():                            # what an application serving such bursts all day might do on top
            quiet = [run(sessions, True) for _ in range(3)]
        dt_quiet = sorted(r[0] for r in quiet)[1]
        untouched = [run(sessions, True, pause_collector=False) for _ in range(3)]
        dt_untouched = sorted(r[0] for r in untouched)[1]
        n_un = min(32, sessions)
        run(4, False)
        dt_un, ok_un, _ = run(n_un, False)
        row = {"workload": "%d concurrent perform_secure_comparison sessions, l=%d, %s + %s" % (sessions, l, pname, dname),
               "value": sessions / dt, "unit": "comparisons/s", "seconds": dt, "sessions": sessions,
               "value_quiet_collector": sessions / dt_quiet, "value_collector_untouched": sessions / dt_untouched,
               "batched_calls": {"initiator": runs[-1][2][0]["calls"], "keyholder": runs[-1][2][1]["calls"], "largest_batch": runs[-1][2][0]["largest"]},
               "seconds_in_batched_calls": {"initiator": {k: round(v, 4) for k, v in runs[-1][2][0]["seconds"].items()},
                                            "keyholder": {k: round(v, 4) for k, v in runs[-1][2][1]["seconds"].items()}},
               "uncoalesced": {"value": n_un / dt_un, "sessions": n_un}, "correct": all(r[1] for r in runs + quiet + untouched) and ok_un}
        if cpu_run is not None:
            try:
                cb = cpu_run(pname, dname)
                row["cpu_oracle"] = {"value": cb["value"], "cores": cb["cores"], "arith": cb["arith"], "sample": cb["count"]}
                row["ratio_to_cpu_oracle"] = row["value"] / cb["value"]
                row["ratio_to_cpu_oracle_quiet_collector"] = row["value_quiet_collector"] / cb["value"]
            except Exception as exc:  # pragma: no cover
                row["cpu_oracle"] = {"error": str(exc)[:200]}
        bob_p.shut_down()
        out.append(row)
    return {"shapes": out,
            "note": "wall clock of asyncio.run over all sessions of both players in one process and event loop (Python object handling included), median of 3; "
                    "every session draws its own randomness and sends / receives its own four messages; value: the library's default -- CPython's cyclic garbage "
                    "collector paused from the first session of a burst until the last has left, 0.5 s at most (coalesce._CollectorPause; "
                    "Initiator / KeyHolder.coalesce_pause_collector_s); value_collector_untouched: the same runs with that set to 0; value_quiet_collector: "
                    "inside coalesce.quiet_collector() as well (the interpreter's existing objects frozen out of the collector's passes, generation-0 "
                    "threshold raised: an application's choice); cpu_oracle = oracle.compare on the box's host cores "
                    "(same key sizes and l, multiprocessing); informational, never `value`"}


def workload_name(B, l, pbits, dbits):
    cfg = {(65536, 32, 2048): "BASELINE configs[2]", (4096, 16, 2048): "BASELINE configs[1]", (131072, 32, 2048): "per-GPU share of BASELINE configs[3]",
           (32768, 64, 3072): "per-GPU share of BASELINE configs[4]"}.get((B, l, pbits), "custom")
    return "batch %d comparisons per GPU, l=%d, %d-bit Paillier + %s DGK (%s)" % (B, l, pbits, "%d-bit" % dbits if dbits else "default", cfg)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="comparisons per GPU per step")
    ap.add_argument("--l", type=int, default=32)
    ap.add_argument("--pbits", type=int, default=2048)
    ap.add_argument("--dgk", default="", help="DGK key fixture (tests/golden/keys.json); default dgk_<pbits>_l<l>")
    ap.add_argument("--rbits", type=int, default=400)
    ap.add_argument("--fb-window", type=int, default=DEFAULT_FB_WINDOW, help="window of the fixed-base table for h (2^w rows of 288 B per window: 0.66 GB at w = 16, 6.2 GB at w = 20, 82 GB at w = 24, HBM-resident, shared by the contexts of a GPU)")
    ap.add_argument("--no-crt", action="store_true")
    ap.add_argument("--no-shuffle", action="store_true", help="leave the step-4i permutation out (do_shuffle=False)")
    ap.add_argument("--streams", type=int, default=0, help="concurrent shards per GPU (one library context, HIP stream and host thread each); 1 = a single stream; "
                    "0 = automatic: 2 once every shard's widest launches still fill the chip (128 comparisons of 2048-bit keys per CU and shard, scaled by the square of the key size)")
    ap.add_argument("--side-stream", type=int, default=-1, help="run the randomizer exponentiations of a step on a second library context and stream per shard, "
                    "concurrently with the protocol's critical path (batch._AheadOfTime): 1 on, 0 off, -1 automatic (on while the batch leaves most wave slots empty: up to 32 comparisons per CU; measured +15 % at 4096, -2 % at 16384 on 256 CUs)")
    ap.add_argument("--side-fork", type=int, default=1, help="fork mode (sc_ctx_set_fork_mode) of the second contexts: 0 = the halves of the key holder's CRT in sequence, 1 = automatic (the q-side on a second stream of that context for small batches, and for large ones when the context has the chip to itself), 2 = always side by side")
    ap.add_argument("--fork-mode", type=int, default=1, help="fork mode of the shard contexts (sc_ctx_set_fork_mode): 0 never, 1 automatic (small batches), 2 always")
    ap.add_argument("--latency-mode", type=int, default=1, help="small-batch kernel policy (sc_ctx_set_latency_mode): 0 never, 1 automatic, 2 always")
    ap.add_argument("--onelane-mode", type=int, default=1, help="large-batch kernel policy for the 1024-bit primes (sc_ctx_set_onelane_mode): 0 never, 1 automatic, 2 always")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0)
    ap.add_argument("--no-extras", action="store_true", help="skip the informational legs (window sensitivity, online phase, PCIe-inclusive)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the compact sub-lines of BASELINE configs[1] and the configs[4] per-GPU share")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even for one rank (exercises the RCCL path)")
    ap.add_argument("--c-abi-gather", action="store_true", help="after the timed region, repeat the reassembly through the C ABI's own RCCL communicator "
                    "(sc_comm_init / sc_allgather) and report whether it equals torch.distributed's gather")
    return ap.parse_args(argv)


class GpuRuntime:
    """Everything bench.py's control flow asks of the device side, in one place: the MI355X through torch's HIP runtime and the
    library's engines.  The CPU test tier drives the SAME main() / measure() with a stand-in (tests/_bench_dry_run.py: gloo instead
    of RCCL, wall-clock events, the test-only engine) so that the N > 1 path -- rank set-up, the per-step gather into a persistent
    array, the rank check, rank 0's line -- is exercised before the first multi-GPU run; this class is the only one the product uses."""

    backend = "nccl"       # RCCL

    def __init__(self, torch) -> None:
        self.torch = torch

    def check_device(self, rank: int, local_rank: int) -> None:
        t = self.torch
        if t.cuda.device_count() <= local_rank:
            raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but this node shows {t.cuda.device_count()} GPU(s) (no CPU fallback)")
        if not t.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (no CPU fallback)")
        t.cuda.set_device(local_rank)

    def dist_device(self, local_rank: int):
        return self.torch.device("cuda", local_rank)

    def synchronize(self) -> None:
        self.torch.cuda.synchronize()

    def event(self):
        return self.torch.cuda.Event(enable_timing=True)

    def stream(self):
        return self.torch.cuda.Stream()

    def current_device(self) -> int:
        return self.torch.cuda.current_device()

    def empty_cache(self) -> None:
        self.torch.cuda.empty_cache()

    def cu_count(self, eng) -> int:
        return self.torch.cuda.get_device_properties(eng.device).multi_processor_count

    def free_memory(self, eng) -> int | None:
        """Free device bytes right now (every rank of a node asks its own GPU before it builds its tables)."""
        free, _total = self.torch.cuda.mem_get_info(eng.device)
        return int(free)

    def default_engine(self):
        from protocols.secure_comparison_amd.schemes import default_engine

        return default_engine()

    def new_engine(self):
        from protocols.secure_comparison_amd.engine import Engine

        return Engine()


class SharedGpuRehearsal(GpuRuntime):
    """SC_BENCH_SHARE_GPU=1: every rank on GPU 0, gloo as the collective (it carries device tensors).  RCCL refuses two ranks on one
    device, and the builder's boxes have one GPU: this is how the N > 1 control flow -- rank set-up, the per-step gather into the
    persistent array, the rank check, the per-rank rows, rank 0's line -- meets the real engines and kernels before the first
    multi-GPU run.  The line says so (`rehearsal`); it is never a scaling measurement."""

    backend = "gloo"
    rehearsal = "every rank on GPU 0, gloo instead of RCCL (SC_BENCH_SHARE_GPU=1): a rehearsal of the N > 1 control flow on real kernels, not a scaling measurement"

    def check_device(self, rank: int, local_rank: int) -> None:
        if not self.torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (no CPU fallback)")
        self.torch.cuda.set_device(0)

    def dist_device(self, local_rank: int):
        return None


def main(argv=None, runtime=None, script: str | None = None) -> None:
    """runtime / script: the CPU dry run of the test tier passes its stand-in for GpuRuntime and its own path (the ranks a
    `--gpus N` parent starts must be the same program)."""
    args = parse_args(argv)
    from protocols.secure_comparison_amd import launcher

    # ---- ranks: start them ourselves, or check the ones a launcher started.  Nothing above this line touches the GPU.
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and launcher.rank_env() is None:
        sys.exit(launcher.spawn_ranks(script or os.path.abspath(__file__), list(sys.argv[1:] if argv is None else argv), args.gpus,
                                      need_gpus=runtime is None and not os.environ.get("SC_BENCH_SHARE_GPU")))
    rank, local_rank, world = launcher.expect_world(args.gpus)

    import torch

    rt = runtime if runtime is not None else (SharedGpuRehearsal(torch) if os.environ.get("SC_BENCH_SHARE_GPU") else GpuRuntime(torch))
    rt.check_device(rank, local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        import datetime

        # a rank that never arrives must not hang the others for torch's default half hour
        dist.init_process_group(rt.backend, rank=rank, world_size=world, device_id=rt.dist_device(local_rank),
                                timeout=datetime.timedelta(seconds=int(os.environ.get("SC_AMD_RENDEZVOUS_TIMEOUT_S", launcher.RENDEZVOUS_TIMEOUT_S))))
    line = measure(args, torch, dist, rank, world, full=True, rt=rt)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def measure(args, torch, dist, rank: int, world: int, full: bool, rt=None):
    """One configuration: set-up, warm-up, the timed steps, and (rank 0) the line's content.  full: the headline run with every
    diagnostic and informational leg; otherwise a compact sub-line (value, ms per step, whole-step fraction) of another BASELINE shape
    measured by the same code path."""
    from protocols.secure_comparison_amd import DGK, Paillier, launcher
    from protocols.secure_comparison_amd.batch import (BatchDraws, ConcurrentShards, PartySet, boot_pools, secure_comparison_batch,
                                                       split_draws)
    from protocols.secure_comparison_amd.distributed import shard_bounds

    rt = rt if rt is not None else GpuRuntime(torch)
    keys = json.load(open(KEYS))
    l, B = args.l, args.batch
    pj = keys[f"paillier_{args.pbits}"]
    dname = args.dgk or f"dgk_{args.pbits}_l{l}"
    dj = keys[dname]
    dbits = (int(dj["p"], 16) * int(dj["q"], 16)).bit_length()
    p, q = int(pj["p"], 16), int(pj["q"], 16)
    H = lambda name: int(dj[name], 16)  # noqa: E731
    use_crt = not args.no_crt
    eng = rt.default_engine() if full else rt.new_engine()
    cus = rt.cu_count(eng)
    # Two concurrent shards pay once each shard's widest launches still fill the chip by themselves: a pair launch holds 16 items per
    # wave and 8 waves per CU, so 128 comparisons of 2048-bit keys per CU and shard (256 CUs: from 32768 comparisons per GPU, +1.4 %
    # there, +5 % at 65536, +7 % at the 3072-bit configs[4] share of 32768); the weight of a comparison grows with the square of the
    # key size.  The randomizers move to a second context while the batch leaves most wave slots empty (up to 32 comparisons per CU).
    ns = max(1, min(args.streams, B)) if args.streams > 0 else (2 if B * (args.pbits / 2048.0) ** 2 >= 128 * cus else 1)
    ns = launcher.host_threads_per_rank(world, ns)      # shard threads of all ranks together stay within the node's cores
    use_side = bool(args.side_stream) if args.side_stream >= 0 else (B <= 32 * cus)
    engines = [eng] + [rt.new_engine() for _ in range(1, ns)]
    side_engines = [rt.new_engine() for _ in range(ns)] if use_side else []
    for e_ in engines:
        e_.set_fork_mode(args.fork_mode)
    for e_ in engines + side_engines:
        e_.set_latency_mode(args.latency_mode)
        e_.set_onelane_mode(args.onelane_mode)
    for e_ in side_engines:
        # Bob's three Paillier randomizers are the second context's longest job and follow rho_z^N on its stream: with their p- and
        # q-halves side by side (the same kernels: one copy of the code per CU) they are done before the critical path needs them,
        # 113.7 k/s at configs[1] with every step within 1 % -- in sequence they were the step's longest chain whenever steps 2 .. 4i
        # got faster (88 .. 103 k/s depending on which launch reached the chip first, profiles/r04_cfg1_second_context_fork.txt)
        e_.set_fork_mode(args.side_fork)

    def build_parties(window: int) -> tuple[list[PartySet], float, int]:
        """Both parties' scheme objects per shard context.  The fixed-base tables are built once (first context) and shared
        read-only by the others; the key holder, who randomizes through CRT, has no table for h mod n at all."""
        sets, build_s, table_bytes = [], 0.0, 0
        for i, e_i in enumerate(engines):
            bob_p = Paillier(p * q, p, q, engine=e_i, use_crt=use_crt)
            bob_d = DGK(H("p") * H("q"), H("g"), H("h"), H("u"), dj["t"], H("p"), H("q"), H("v_p"), H("v_q"), engine=e_i,
                        randomizer_bits=args.rbits, fixed_base_window=window, use_crt=use_crt)
            alice_d = bob_d.public_copy()
            if i > 0:
                bob_d.share_tables_from(sets[0].bob_dgk)
                alice_d.share_tables_from(sets[0].alice_dgk)
            alice_d.prepare(), bob_d.prepare()    # key objects + tables: untimed set-up, like key generation ("excluding table build")
            alice_pai = bob_p.public_copy()
            _ = bob_p.key, alice_pai.key          # Paillier key objects (moduli, exponents, CRT constants): set-up as well
            if i == 0:
                build_s = alice_d.table_build_s + bob_d.table_build_s
                table_bytes = alice_d.table_bytes() + bob_d.table_bytes()
            sets.append(PartySet(alice_pai, alice_d, bob_p, bob_d, rt.stream()))
        for i, e_s in enumerate(side_engines):      # the second context of every shard: same keys, the first context's tables
            bob_p = Paillier(p * q, p, q, engine=e_s, use_crt=use_crt)
            bob_d = DGK(H("p") * H("q"), H("g"), H("h"), H("u"), dj["t"], H("p"), H("q"), H("v_p"), H("v_q"), engine=e_s,
                        randomizer_bits=args.rbits, fixed_base_window=window, use_crt=use_crt)
            alice_d = bob_d.public_copy()
            bob_d.share_tables_from(sets[0].bob_dgk)
            alice_d.share_tables_from(sets[0].alice_dgk)
            alice_d.prepare(), bob_d.prepare()
            alice_pai = bob_p.public_copy()
            _ = bob_p.key, alice_pai.key
            sets[i].side = PartySet(alice_pai, alice_d, bob_p, bob_d, rt.stream())
        return sets, build_s, table_bytes

    requested_window = args.fb_window
    args.fb_window, window_note = choose_fixed_base_window(requested_window, rt.free_memory(eng) if hasattr(rt, "free_memory") else None, args.rbits, dbits,
                                                           dj["t"], use_crt)
    if window_note and rank == 0:
        print("bench.py: " + window_note, file=sys.stderr, flush=True)
    parties, table_build_s, table_bytes = build_parties(args.fb_window)
    alice_p, alice_d, bob_p, bob_d = parties[0].alice_paillier, parties[0].alice_dgk, parties[0].bob_paillier, parties[0].bob_dgk
    x, y, x_enc, y_enc, draws = synth_inputs(eng, l, alice_p, bob_p, bob_d, B, args.rbits, seed=rank, shuffle=not args.no_shuffle)

    # ---- concurrent shards (batch.ConcurrentShards): the batch is cut into `ns` contiguous shards once, outside the timed
    # region (a caller that produces its inputs per shard pays nothing; cutting a resident batch is one ~1.5 GB device copy)
    shard_inputs = None
    if ns > 1:
        bounds = [shard_bounds(B, i, ns) for i in range(ns)]
        shard_inputs = [(x_enc[a:b].contiguous(), y_enc[a:b].contiguous(), d) for (a, b), d in zip(bounds, split_draws(draws, bounds))]
        rt.synchronize()

    # persistent result arrays: every shard writes its rows into its block of `result_buf` (no concatenation pass inside a step),
    # and the per-step all-gather fills the same `gather_buf` every time
    result_buf = torch.empty((B, 2 * alice_p.mod_n.nwords), dtype=torch.int32, device=eng.device)
    gather_buf = None if dist is None else torch.empty((world * B, result_buf.shape[1]), dtype=torch.int32, device=eng.device)

    def make_step(party_sets):
        """(step function, closer): the closer ends the shard threads and hands the chip back to single-stream policies."""
        if ns == 1:
            ps = party_sets[0]
            return (lambda: secure_comparison_batch(x_enc, y_enc, l, ps.alice_paillier, ps.alice_dgk, ps.bob_paillier, ps.bob_dgk, draws,
                                                    randomize=True, side=ps.side, out=result_buf)), (lambda: None)
        runner = ConcurrentShards(party_sets)
        return (lambda: runner.run(shard_inputs, l, randomize=True, out=result_buf)), runner.close

    step, close_step = make_step(parties)

    def gather(res):
        if dist is None:
            return res
        dist.all_gather_into_tensor(gather_buf, res)
        return gather_buf

    def timed(fn, steps, marks=None):
        """`steps` steps between two barrier + synchronize brackets; the time is the maximum over the ranks.  marks: a list that
        receives one event per step boundary (recorded on the caller's stream, which waits for the shard streams at the end of
        every step: no synchronisation is added) -- the per-step spread is read from them afterwards."""
        if dist is not None:
            dist.barrier()
        rt.synchronize()
        t0 = time.perf_counter()
        if marks is not None:
            marks.append(rt.event())
            marks[-1].record()
        for _ in range(steps):
            r_ = gather(fn())
            if marks is not None:
                marks.append(rt.event())
                marks[-1].record()
        rt.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=eng.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, r_

    def dominant_launch(reps=3):
        """Alice's rho^N mod N^2 for the whole batch alone on the chip (k_pvm<4,18> + the assembly launch): seconds per launch by HIP
        events on the stream the library launches on, executed multiply-adds per launch, the pure multiply-add probe's rate."""
        ev0, ev1 = rt.event(), rt.event()
        alice_p.randomize_batch(x_enc, draws.rho_z)
        rt.synchronize()
        before = eng.mac_counter()
        ev0.record()
        for _ in range(reps):
            alice_p.randomize_batch(x_enc, draws.rho_z)
        ev1.record()
        rt.synchronize()
        return ev0.elapsed_time(ev1) * 1e-3 / reps, (eng.mac_counter() - before) / reps, eng.peak_probe()

    def clock_ghz():
        """Engine clock the dominant pair launch holds, from the stamping twin of the kernel (sc_clock_probe; None where the batch
        does not take the (4,18) modulus-multiple launch)."""
        try:
            return eng.clock_probe(alice_p.key, draws.rho_z)[0]
        except Exception:
            return None

    # ---- diagnostics BEFORE anything else has loaded the chip (rank 0 of the headline run): the dominant launch, the probe and the
    # in-kernel clock -- repeated after the timed loop, so that the line itself says whether the chip's clock state moved
    diag_before = None
    if full:       # every rank: with N GPUs the slowest one sets a weak-scaling step, and the line must be able to say which it was
        ls_, _, pk_ = dominant_launch(reps=2)
        diag_before = {"launch_ms": ls_ * 1e3, "probe_peak": pk_ / 1e12, "clock_ghz": clock_ghz()}

    res = None
    for _ in range(args.warmup):
        res = gather(step())
    rt.synchronize()
    # parity spot check outside the timed region: decrypt the results on the GPU and compare with x <= y
    expect = (x <= y).to(torch.int32)

    def all_correct(full):
        dec = bob_p.decrypt_raw_batch((full[rank * B:(rank + 1) * B] if world > 1 else full).contiguous())
        return bool(((dec[:, 0] == expect) & (dec[:, 1:] == 0).all(dim=1)).all().item())

    if res is not None and not all_correct(res):
        raise SystemExit("bench.py: decrypted results differ from x <= y")
    rccl_ranks, devices = 1, [rt.current_device()]
    if dist is not None:   # prove the collective sees every rank, and where the ranks sit
        ones = torch.ones(1, dtype=torch.int64, device=eng.device)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
        ids = torch.zeros(world, dtype=torch.int64, device=eng.device)
        dist.all_gather_into_tensor(ids, torch.tensor([rt.current_device()], dtype=torch.int64, device=eng.device))
        devices = ids.tolist()
        if rccl_ranks != world:
            raise SystemExit(f"bench.py: the all-reduce saw {rccl_ranks} ranks, expected {world}")
    [e_.mac_counter(reset=True) for e_ in engines + side_engines]
    marks = []
    elapsed, res = timed(step, args.steps, marks)
    executed_macs = sum(e_.mac_counter() for e_ in engines + side_engines)
    value = world * B * args.steps / elapsed
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1))
    if not all_correct(res):
        raise SystemExit("bench.py: decrypted results of the timed steps differ from x <= y")
    # ---- per-rank view (outside the timed region): this rank's step WITHOUT the collective (what its GPU does alone), its dominant
    # launch, probe and clock before / after -- gathered into rank 0's line with one small all-gather
    per_rank, weak_eff = None, None
    if full:
        solo_marks = [rt.event() for _ in range(3)]
        solo_marks[0].record()
        for i_ in range(2):
            step()
            solo_marks[i_ + 1].record()
        rt.synchronize()
        solo_ms = min(solo_marks[i_].elapsed_time(solo_marks[i_ + 1]) for i_ in range(2))
        la_, _, pa_ = dominant_launch(reps=2)
        ck_ = clock_ghz()
        mine = [float(rank), float(rt.current_device()), step_ms[len(step_ms) // 2], step_ms[0], step_ms[-1], solo_ms, diag_before["launch_ms"], la_ * 1e3,
                diag_before["clock_ghz"] or 0.0, ck_ or 0.0, diag_before["probe_peak"], pa_ / 1e12, float(table_bytes), float(args.fb_window)]
        rows_ = [mine]
        if dist is not None:
            dev_ = eng.device if rt.backend == "nccl" else "cpu"
            buf_ = torch.zeros(world * len(mine), dtype=torch.float64, device=dev_)       # (flat: gloo's gather wants the concatenated form)
            dist.all_gather_into_tensor(buf_, torch.tensor(mine, dtype=torch.float64, device=dev_))
            rows_ = buf_.reshape(world, len(mine)).cpu().tolist()
        names_ = ("rank", "device", "step_ms_median", "step_ms_min", "step_ms_max", "solo_step_ms", "launch_ms_before", "launch_ms_after", "clock_ghz_before",
                  "clock_ghz_after", "probe_peak_before", "probe_peak_after", "fixed_base_table_bytes", "fixed_base_window")
        per_rank = [dict(zip(names_, r_)) for r_ in rows_]
        for pr_ in per_rank:
            pr_["rank"], pr_["device"], pr_["fixed_base_window"] = int(pr_["rank"]), int(pr_["device"]), int(pr_["fixed_base_window"])
            pr_["fixed_base_table_bytes"] = int(pr_["fixed_base_table_bytes"])
        slowest = max(pr_["solo_step_ms"] for pr_ in per_rank)
        weak_eff = value / (world * B / (slowest * 1e-3))
    close_step()      # the single-stream measurements below run with the policies of a context that has the chip to itself
    nominal_peak = cus * 64 * MAX_CLOCK_HZ   # 4 SIMDs x 16 lanes per CU, one multiply-add per lane and cycle
    if not full:      # a compact sub-line of another BASELINE shape
        sub = None
        if rank == 0:
            sub = {"workload": workload_name(B, l, args.pbits, dbits), "value": value, "unit": "comparisons/s", "ms_per_step": elapsed / args.steps * 1e3,
                   "steps": args.steps, "streams_per_gpu": ns, "randomizers_on_side_stream": use_side, "dgk_key": dname,
                   "whole_step_frac": executed_macs / (elapsed * nominal_peak), "all_rows_decrypt_to_x_le_y": True}
            if not args.no_cpu_baseline:     # like the headline: rows of the timed batch under the oracle, bit for bit (round 4 held these to the decrypt property)
                try:
                    mine_ = (res[rank * B:(rank + 1) * B] if world > 1 else res).contiguous()
                    sub["oracle"] = oracle_rows_check(eng, 64, B, l, x_enc, y_enc, draws, mine_, f"paillier_{args.pbits}", dname, args.rbits)
                except Exception as exc:  # pragma: no cover
                    sub["oracle"] = {"error": str(exc)[:200]}
                if sub["oracle"].get("equal_to_oracle") != sub["oracle"].get("rows"):
                    raise SystemExit(f"bench.py: the CPU oracle and the GPU disagree on sampled rows of {sub['workload']}: {sub['oracle']}")
        del parties, step, close_step
        for e_ in engines + side_engines:
            e_.close()
        rt.empty_cache()
        return sub
    # outside the timed region: the same reassembly through the C ABI's own RCCL communicator (sc_comm_init / sc_allgather,
    # what a host without torch would call), compared with torch.distributed's gather.  Opt-in (--c-abi-gather: a second
    # communicator's rendezvous is not something the scaling run should depend on); reported, never fatal.
    c_abi_gather = None
    if dist is not None and args.c_abi_gather:
        try:
            from protocols.secure_comparison_amd.distributed import comm_init_from_torch

            comm_init_from_torch(eng)
            mine = res[rank * B:(rank + 1) * B].contiguous() if world > 1 else res.contiguous()
            again = eng.allgather(mine)
            rt.synchronize()
            ok_t = torch.tensor([int(torch.equal(again, res))], dtype=torch.int64, device=eng.device)
            dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
            c_abi_gather = {"equal_to_torch_gather": bool(ok_t.item()), "ranks": world}
            eng.comm_destroy()
        except Exception as exc:  # pragma: no cover
            c_abi_gather = {"error": str(exc)[:200]}

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel: Alice's Paillier randomizer rho^N mod N^2 (k_pvm<4,18> + assembly launch),
        # timed with HIP events on the stream the library launches on (torch's current stream here)
        launch_s, launch_exec_macs, peak = dominant_launch(reps=3)
        diag_after = {"launch_ms": launch_s * 1e3, "probe_peak": peak / 1e12, "clock_ghz": clock_ghz()}
        ev0, ev1 = rt.event(), rt.event()
        s32 = 2 * args.pbits // 32
        alg_macs = B * (args.pbits + -(-args.pbits // 5) + 30 + 1) * (2 * s32 * s32 + s32)
        alg_bytes_launch = B * (args.pbits // 8 + 2 * (2 * args.pbits // 8))      # rho in, ciphertext in, ciphertext out
        # ---- modexp/s for the two canonical shapes of SURVEY 8(d): P = Paillier randomizer, D = DGK fixed-base randomizer
        d_exps = draws.r_alice_dgk.reshape((l + 1) * B, -1)
        alice_d.randomize_batch(None, d_exps)
        rt.synchronize()
        ev0.record()
        alice_d.randomize_batch(None, d_exps)
        ev1.record()
        rt.synchronize()
        d_rate = d_exps.shape[0] / (ev0.elapsed_time(ev1) * 1e-3)
        lit = literal_macs_per_comparison(l, args.pbits, dbits, args.rbits)
        abytes = algorithmic_bytes_per_comparison(l, args.pbits, dbits, args.rbits)
        # HBM traffic of the dominant launch from the committed PMC pass (rocprofv3 cannot run inside this process)
        traffic = None
        for name in ("r05_dominant_kernel_traffic.json", "r04_dominant_kernel_traffic.json", "r03_dominant_kernel_traffic.json", "r02_dominant_kernel_traffic.json", "r01_dominant_kernel_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tpath) and B == 65536 and args.pbits == 2048 and l == 32 and use_crt:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
                break
        out = {
            "metric": "secure comparisons/sec (l=%d, %d-bit keys)" % (l, args.pbits), "value": value, "unit": "comparisons/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64 (29-bit limbs held in u32, 32x32+64->64 multiply-accumulate)",
            "data": "synthetic", "rccl_ranks": rccl_ranks, "rank_devices": devices, "c_abi_gather": c_abi_gather,
            **({"rehearsal": rt.rehearsal} if getattr(rt, "rehearsal", None) else {}),
            "step_ms": {"min": step_ms[0], "median": step_ms[len(step_ms) // 2], "max": step_ms[-1],
                        "note": "per-step device time between events on the caller's stream inside the timed region (no synchronisation added)"},
            "per_rank": per_rank,
            "weak_scaling_efficiency": weak_eff,
            "per_rank_note": "one row per rank, gathered outside the timed region: its steps inside the timed loop (median / min / max, collective included), "
                             "its step WITHOUT the collective (solo_step_ms, best of two), its dominant launch, in-kernel clock and multiply-add probe before the "
                             "warm-up and after the loop, its table; weak_scaling_efficiency = value / (N x B / slowest rank's solo step): what the collective "
                             "and the slowest GPU cost together",
            "config": {"workload": workload_name(B, l, args.pbits, dbits),
                       "batch_per_gpu": B, "l": l, "paillier_bits": args.pbits, "dgk_bits": dbits, "dgk_key": dname, "dgk_randomizer_bits": args.rbits,
                       "fixed_base_window": args.fb_window, "fixed_base_window_requested": requested_window, "fixed_base_window_note": window_note,
                       "keyholder_crt": use_crt, "shuffle_4i": not args.no_shuffle,
                       "parallelism": "shard%d" % world, "streams_per_gpu": ns, "randomizers_on_side_stream": use_side,
                       "fixed_base_table_bytes": table_bytes, "table_build_s": table_build_s,
                       "table_note": "device bytes of Alice's table for h mod n plus the key holder's CRT tables for h mod p, h mod q, built once per GPU "
                                     "(untimed set-up, SURVEY 8(d)) and shared read-only by the %d shard context(s)" % ns},
            "roofline": {"bound": "valu-int",
                         "bound_note": "v_mad_u64_u32 issue rate; neither HBM nor MFMA bounds this path (SURVEY 8(d)); the HBM view is in roofline_hbm",
                         "kernel": "k_pvm<4,18>: Paillier randomizer rho^N mod N^2 for B items, pair arithmetic modulo N (one launch) "
                                   "followed by the k_vm<8,18> launch that assembles w0 + w1 N and multiplies into the ciphertext",
                         "achieved": launch_exec_macs / launch_s / 1e12, "peak": nominal_peak / 1e12, "unit": "T MAC/s (v_mad_u64_u32 lane operations executed)",
                         "frac": launch_exec_macs / launch_s / nominal_peak,
                         "peak_note": "peak = CUs x 4 SIMDs x 16 lanes x 2.4 GHz (MI355X peak engine clock): a wave64 v_mad_u64_u32 occupies its SIMD for 4 cycles; "
                                      "probe_peak = sc_peak_probe, a pure v_mad_u64_u32 stream measured on this box",
                         "probe_peak": peak / 1e12, "frac_of_probe": launch_exec_macs / launch_s / peak,
                         "executed_macs_per_launch": launch_exec_macs,
                         "literal_opmix_ratio": alg_macs / launch_s / nominal_peak,
                         "literal_note": "SURVEY 8(d)'s LITERAL op mix (32-bit-limb CIOS, window-5 modexp mod N^2) / launch time / peak: "
                                         "algorithmic savings (pair arithmetic, symmetric squaring, sliding window 6) push it above frac, and above 1",
                         "literal_macs_per_launch": alg_macs,
                         "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes_launch,
                         "traffic_note": "HBM bytes moved per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from the committed PMC passes (profiles/; gfx950 tallies each 128-B line request at 64 B, "
                                         "calibrated on this library's limb rows in profiles/r01_traffic_calibration.json); it exceeds the algorithmic bytes because each item's "
                                         "window table lives in an HBM scratch arena (DESIGN.md 4)",
                         "launch_ms": launch_s * 1e3,
                         "launch_ms_before": diag_before["launch_ms"], "launch_ms_after": diag_after["launch_ms"],
                         "probe_peak_before": diag_before["probe_peak"], "probe_peak_after": diag_after["probe_peak"],
                         "clock_ghz_before": diag_before["clock_ghz"], "clock_ghz_after": diag_after["clock_ghz"],
                         "before_after_note": "the dominant launch, the pure multiply-add probe and the in-kernel clock (sc_clock_probe: s_memtime / s_memrealtime stamps of a twin "
                                              "of the kernel, never a timed launch) measured BEFORE the warm-up and AFTER the timed loop: a line whose value moved "
                                              "while these did not has changed code, one where they moved together has a chip in another clock state"},
            "policy": policy_of(eng),
            "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
            "modexp_per_s": {"P": B / launch_s, "D": d_rate,
                             "shapes": "P: %d-bit base ^ %d-bit exponent mod %d-bit (rho^N mod N^2); D: fixed base, %d-bit exponent mod %d-bit (h^r mod n)"
                                       % (args.pbits, args.pbits, 2 * args.pbits, args.rbits, dbits)},
            "roofline_whole_step": {"executed_macs_per_comparison": executed_macs / (B * args.steps),
                                    "achieved": executed_macs / elapsed / 1e12, "peak": nominal_peak / 1e12, "unit": "T MAC/s per GPU (executed)",
                                    "frac": executed_macs / (elapsed * nominal_peak), "frac_of_probe": executed_macs / (elapsed * peak),
                                    "literal_macs_per_comparison": lit, "literal_opmix_ratio": lit * value / world / nominal_peak},
            "roofline_hbm": {"bound": "hbm", "achieved": abytes * value / world / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": abytes * value / world / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_comparison": abytes},
        }
        cpu_run, py_cpu, cores_cpu = None, None, 0
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is reported at N = 1 only
            cores = min(os.cpu_count() or 1, 16)
            sample = min(B, args.cpu_sample or 64 * cores)
            py = cpu_interpreter()
            py_cpu, cores_cpu = py, cores
            try:
                stride = max(1, B // sample)
                idx = [i * stride for i in range(sample)]
                with tempfile.TemporaryDirectory() as td:
                    path = os.path.join(td, "sample.json")
                    export_sample(path, eng, idx, l, x_enc, y_enc, draws, res)

                    def cpu_run(interp, count, procs, pname=f"paillier_{args.pbits}", dgk_name=dname, sample_path=path):
                        return run_cpu_oracle(interp, count, procs, pname, dgk_name, args.rbits, sample_path)

                    cb = cpu_run(py, sample, cores)
                    out["cpu_baseline"] = {"value": cb["value"], "unit": "comparisons/s", "cores": cb["cores"], "kind": "port",
                                           "sample": "%d comparisons of the SAME seeded batch (every %d-th row: inputs and all random draws downloaded from the GPU), "
                                                     "oracle.compare with %s over %d processes; %d/%d results equal the GPU's bit for bit"
                                                     % (cb["count"], stride, cb["arith"], cb["cores"], cb["match_gpu"], cb["count"])}
                    one = cpu_run(py, 8, 1)                       # single-core figure
                    out["cpu_baseline"]["single_core"] = {"value": one["value"], "arith": one["arith"], "sample": one["count"]}
                    c0 = cpu_run(py, 32, 1, "paillier_1024", "dgk_1024_l16", None)      # configs[0]'s shape on one core (latency_single below)
                    out["cpu_baseline"]["configs0_single_core_ms"] = 1e3 / c0["value"]
                    if py != sys.executable:                      # and the pure-Python-int path of the default interpreter
                        pure = cpu_run(sys.executable, 2 * cores, cores)
                        out["cpu_baseline"]["python_int"] = {"value": pure["value"], "cores": pure["cores"], "arith": pure["arith"], "sample": pure["count"]}
                    if cb["match_gpu"] != cb["count"]:
                        raise SystemExit("bench.py: the CPU oracle and the GPU disagree on the sampled comparisons")
            except SystemExit:
                raise
            except Exception as exc:  # pragma: no cover
                out["cpu_baseline"] = {"value": None, "unit": "comparisons/s", "cores": cores, "kind": "port", "sample": f"failed: {exc}"}
        if not args.no_extras and world == 1:
            def guarded(name, leg):
                """An informational leg must never cost the line its headline: a failure is recorded under the leg's name."""
                try:
                    out[name] = leg()
                except Exception as exc:  # pragma: no cover
                    out[name] = {"error": repr(exc)[:300]}

            def leg_window_sensitivity():
                # ---- how much of `value` hangs on the 6 GB window-20 table: the same step with smaller fixed-base windows
                sens = {str(args.fb_window): {"value": value, "table_bytes": table_bytes, "table_build_s": table_build_s}}
                for w in (8, 16, 20):
                    if w == args.fb_window:
                        continue
                    ps_w, bs_w, tb_w = build_parties(w)
                    step_w, close_w = make_step(ps_w)
                    step_w()
                    dt_w, _ = timed(step_w, 2)
                    close_w()
                    sens[str(w)] = {"value": B * 2 / dt_w, "table_bytes": tb_w, "table_build_s": bs_w}
                    del ps_w, step_w, close_w
                    rt.empty_cache()
                return sens

            def leg_interactive_protocol():
                # ---- the real two-party batch protocol (reported, never `value`): Initiator / KeyHolder
                # .perform_secure_comparison_batch with draws=None over the in-memory transport -- every random input drawn on the
                # device by the library's CSPRNG inside the timed region, messages handed over as device arrays (or, second figure,
                # serialized through one pinned host buffer per message as a real transport would need)
                def one_more_party():      # a further session's context and scheme objects (same keys, the first context's tables)
                    e_x = rt.new_engine()
                    e_x.set_latency_mode(args.latency_mode)
                    e_x.set_onelane_mode(args.onelane_mode)
                    bob_px = Paillier(p * q, p, q, engine=e_x, use_crt=use_crt)
                    bob_dx = DGK(H("p") * H("q"), H("g"), H("h"), H("u"), dj["t"], H("p"), H("q"), H("v_p"), H("v_q"), engine=e_x,
                                 randomizer_bits=args.rbits, fixed_base_window=args.fb_window, use_crt=use_crt)
                    alice_dx = bob_dx.public_copy()
                    bob_dx.share_tables_from(parties[0].bob_dgk)
                    alice_dx.share_tables_from(parties[0].alice_dgk)
                    alice_dx.prepare(), bob_dx.prepare()
                    alice_px = bob_px.public_copy()
                    _ = bob_px.key, alice_px.key
                    return PartySet(alice_px, alice_dx, bob_px, bob_dx, rt.stream())

                return interactive_protocol_leg(torch, eng, parties, ns, shard_inputs, x_enc, y_enc, expect, l, B, value, one_more_party)

            def leg_latency_single():
                # ---- BASELINE configs[0]: the latency of ONE comparison through the product (no CPU path)
                ls_ = latency_single_leg(torch, eng, keys)
                ls_["cpu_oracle_ms"] = out.get("cpu_baseline", {}).get("configs0_single_core_ms")
                return ls_

            def leg_online_phase_only():
                # ---- online phase only (reported, never `value`): randomizers pre-generated into device pools (untimed), as the
                # reference pre-generates them in background workers (boot_randomness_generation, SC/initiator.py:205-210)
                gen = torch.Generator(device=eng.device)
                gen.manual_seed(1234)
                boot_pools(B, l, alice_p, alice_d, bob_p, bob_d, source="torch", generator=gen)
                rt.synchronize()
                to = time.perf_counter()
                ro = secure_comparison_batch(x_enc, y_enc, l, alice_p, alice_d, bob_p, bob_d, draws, randomize="pool")
                rt.synchronize()
                online_s = time.perf_counter() - to
                dec_o = bob_p.decrypt_raw_batch(ro)
                return {"value": B / online_s, "unit": "comparisons/s", "correct": bool((dec_o[:, 0] == expect).all().item()),
                                            "note": "all 4 + 2(l+1) randomizer exponentiations per comparison pre-generated (excluded); informational"}

            def leg_pcie_inclusive():
                # ---- PCIe-inclusive rate (reported at N = 1, never `value`): inputs start in pinned host memory, result returns to the host
                names = ("r", "delta_a", "rhos", "rho_z", "r_bob_dgk", "r_alice_dgk", "rho_zeta_1", "rho_zeta_2", "rho_delta_b")
                host_in = [t.cpu().pin_memory() for t in (x_enc, y_enc) + tuple(getattr(draws, n_) for n_ in names)]
                host_perm = None if draws.permutation is None else draws.permutation.cpu().pin_memory()
                for _ in range(2):                                # the first pass pays for the allocator's first-time hipMallocs
                    rt.synchronize()
                    tp = time.perf_counter()
                    dv = [t.to(eng.device, non_blocking=True) for t in host_in]
                    d2 = BatchDraws(permutation=None if host_perm is None else host_perm.to(eng.device, non_blocking=True),
                                    **{n_: dv[2 + i] for i, n_ in enumerate(names)})
                    r2 = secure_comparison_batch(dv[0], dv[1], l, alice_p, alice_d, bob_p, bob_d, d2, randomize=True).cpu()
                    rt.synchronize()
                    pcie_s = time.perf_counter() - tp
                    del dv, d2
                return {"value": B / pcie_s, "unit": "comparisons/s", "host_bytes_per_comparison":
                                         sum(t.numel() * t.element_size() for t in host_in) / B + r2.shape[-1] * 4}

            def leg_concurrent_sessions():
                # ---- many concurrent SINGLE comparisons on one player pair (the reference's call shape), coalesced into batch launches
                def cpu_shape(pname, dgk_name):
                    return cpu_run(py_cpu, 4 * cores_cpu, cores_cpu, pname, dgk_name, None) if cpu_run is not None else None

                return concurrent_sessions_leg(torch, eng, keys, cpu_run=cpu_shape if cpu_run is not None else None)

            guarded("window_sensitivity", leg_window_sensitivity)
            guarded("interactive_protocol", leg_interactive_protocol)
            guarded("latency_single", leg_latency_single)
            guarded("concurrent_sessions", leg_concurrent_sessions)
            guarded("online_phase_only", leg_online_phase_only)
            guarded("pcie_inclusive", leg_pcie_inclusive)
        if not args.no_other_configs and world == 1 and (B, l, args.pbits) == (65536, 32, 2048):
            # ---- the other BASELINE shapes that fit one GPU, by the same code path, so that they are driver-run numbers too
            import copy

            others = []
            for kw in ({"batch": 4096, "l": 16, "pbits": 2048, "dgk": "dgk_2048_l16", "steps": 10, "warmup": 2},
                       {"batch": 32768, "l": 64, "pbits": 3072, "dgk": "dgk_2048_l64", "steps": 2, "warmup": 1}):
                a2 = copy.copy(args)
                for k_, v_ in kw.items():
                    setattr(a2, k_, v_)
                a2.streams, a2.side_stream = 0, -1
                a2.fb_window = min(args.fb_window, 20)      # other keys: their own tables beside the headline's 82 GB
                try:
                    others.append(measure(a2, torch, None, 0, 1, full=False, rt=rt))
                except Exception as exc:  # pragma: no cover
                    others.append({"workload": workload_name(kw["batch"], kw["l"], kw["pbits"], None), "error": str(exc)[:200]})
            out["other_configs"] = others
    return out


if __name__ == "__main__":
    main()
