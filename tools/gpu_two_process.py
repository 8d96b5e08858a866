"""Alice and Bob as TWO OS PROCESSES on one GPU (round-4 review, "what's missing" 4): the reference's players are separate processes or
hosts (SC/test/integration/test_pool.py:41-73, test/conftest.py:162-198); round 4 measured the byte transport between two threads of
one process only.

    python tools/gpu_two_process.py [--batch 65536] [--l 32] [--pbits 2048] [--batches 6] [--chunks 1] [--window 20] [--sessions 1]

The parent never touches the GPU: it starts the key holder and the initiator as fresh children (this file with --role), each with
its own HIP context, library context and generator; they talk over a Unix socket through communicator.StreamCommunicator in the
wire format of wire.py (public scheme document, then per batch the four byte messages).  Every random input is drawn on the
device by each party's own generator (draws=None).  After the timed batches the initiator ships her last results and the expected
bits to the key holder, who decrypts and answers with the number of correct rows -- she has no secret key to check them herself.
Prints one JSON object."""
import argparse
import asyncio
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--role", default="parent")
    ap.add_argument("--socket", default="")
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--l", type=int, default=32)
    ap.add_argument("--pbits", type=int, default=2048)
    ap.add_argument("--batches", type=int, default=6)
    ap.add_argument("--chunks", type=int, default=1)
    ap.add_argument("--window", type=int, default=20)
    ap.add_argument("--sessions", type=int, default=1)
    return ap.parse_args()


def _keys(args):
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "keys.json")))
    pj, dj = k[f"paillier_{args.pbits}"], k[f"dgk_{args.pbits if args.pbits < 3072 else 2048}_l{args.l}"]
    H = lambda d, name: int(d[name], 16)  # noqa: E731
    return (H(pj, "p"), H(pj, "q")), {n: (H(dj, n) if n != "t" else dj["t"]) for n in ("p", "q", "g", "h", "u", "t", "v_p", "v_q")}


def _engines(args):
    """One library context per session; the first is the process's default context."""
    from protocols.secure_comparison_amd.engine import Engine
    from protocols.secure_comparison_amd.schemes import default_engine

    first = default_engine()
    engines = [first] + [Engine(first.device_index) for _ in range(1, args.sessions)]
    for e in engines:
        e.set_chip_share(args.sessions)
    return engines


def _in_threads(args, session):
    """Run session(i) for every session of this process, each on a thread with its own event loop and stream."""
    import threading

    import torch

    engines, out, errors = _engines(args), [None] * args.sessions, []
    prepared = {}

    def body(i):
        try:
            with torch.cuda.device(engines[i].device), torch.cuda.stream(torch.cuda.Stream()):
                out[i] = asyncio.run(session(i, engines[i], prepared))
        except BaseException as exc:  # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=body, args=(i,)) for i in range(args.sessions)]
    body_first = threads[0]
    body_first.start()                      # the first session builds the fixed-base tables; the others import them
    while "tables" not in prepared and body_first.is_alive():
        time.sleep(0.01)
    [t.start() for t in threads[1:]]
    [t.join() for t in threads]
    if errors:
        raise SystemExit("; ".join(errors)[:1000])
    return out


def _path(args, i):
    return args.socket if i == 0 else f"{args.socket}.{i}"


async def keyholder(args, i, eng, prepared):
    import torch

    from protocols.secure_comparison_amd import DGK, KeyHolder, Paillier, StreamCommunicator, wire

    (p, q), d = _keys(args)
    bob_p = Paillier(p * q, p, q, engine=eng)
    bob_d = DGK(d["p"] * d["q"], d["g"], d["h"], d["u"], d["t"], d["p"], d["q"], d["v_p"], d["v_q"], engine=eng, randomizer_bits=400,
                fixed_base_window=args.window)
    if i > 0:
        bob_d.share_tables_from(prepared["tables"])
    bob_d.prepare()
    if i == 0:
        prepared["tables"] = bob_d
    _ = bob_p.key
    comm = await StreamCommunicator.accept_unix(_path(args, i), alloc=wire.pinned_buffer)       # batch messages land in pinned host memory
    bob = KeyHolder(args.l, comm, "initiator", bob_p, bob_d)
    for _ in range(args.batches + 1):                      # one warm-up batch, then the timed ones
        await bob.perform_secure_comparison_batch()
    res, expect = wire.unpack_many(await comm.recv("initiator", "check"), eng.device, expect=2)
    dec = bob_p.decrypt_raw_batch(res.contiguous())
    ok = int(((dec[:, 0] == expect.reshape(-1).to(torch.int32)) & (dec[:, 1:] == 0).all(dim=1)).sum().item())
    await comm.send("initiator", json.dumps({"ok": ok, "rows": int(res.shape[0])}).encode(), "verdict")
    await comm.close()


_start = None           # (threading.Barrier, [t0]): the initiator's sessions leave their warm-up together


async def initiator(args, i, eng, prepared):
    import torch

    import bench
    from protocols.secure_comparison_amd import DGK, Initiator, Paillier, StreamCommunicator, wire

    (p, q), d = _keys(args)
    per = args.batch // args.sessions
    pub_p = Paillier(p * q, engine=eng)                         # public parts only, as received keys would be
    pub_d = DGK(d["p"] * d["q"], d["g"], d["h"], d["u"], d["t"], engine=eng, randomizer_bits=400, fixed_base_window=args.window)
    if i > 0:
        pub_d.share_tables_from(prepared["tables"])
    pub_d.prepare()
    if i == 0:
        prepared["tables"] = pub_d
    x, y, x_enc, y_enc, _ = bench.synth_inputs(eng, args.l, pub_p, pub_p, pub_d, per, 400, seed=i)
    expect = (x <= y).to(torch.int32)
    comm = await StreamCommunicator.open_unix(_path(args, i), alloc=wire.pinned_buffer, wait_s=300)
    alice = Initiator(args.l, comm, "keyholder", pub_p, pub_d)
    stream = torch.cuda.current_stream()
    await alice.perform_secure_comparison_batch(x_enc, y_enc, engine=eng, chunks=args.chunks)         # warm-up: tables, programs, pinned pools
    stream.synchronize()
    _start[0].wait()
    if i == 0:
        wire.reset_stats()
        _start[1].append(time.perf_counter())
    _start[0].wait()
    t0 = _start[1][0]
    time.sleep(i * 0.36 * args.batch / 65536 / args.sessions)          # a fraction of a batch apart: transfers of one beside launches of the other
    stamps = []
    res = None
    for _ in range(args.batches):
        res = await alice.perform_secure_comparison_batch(x_enc, y_enc, engine=eng, chunks=args.chunks)
        stream.synchronize()
        stamps.append(time.perf_counter() - t0)
    await comm.send("keyholder", wire.pack_many(res.contiguous(), expect), "check")
    verdict = json.loads(bytes(await comm.recv("keyholder", "verdict")).decode())
    await comm.close()
    return stamps, verdict


def initiator_report(args, outs):
    from protocols.secure_comparison_amd import wire

    per = args.batch // args.sessions
    done = sorted(t for stamps, _ in outs for t in stamps)
    total = done[-1]
    n, span = len(done) - args.sessions, done[-1] - done[args.sessions - 1]
    gaps = sorted(b - a for stamps, _ in outs for a, b in zip([0.0] + stamps, stamps))
    print(json.dumps({"two_process": True, "value": per * len(done) / total, "unit": "comparisons/s", "batch": args.batch, "l": args.l,
                      "sessions": args.sessions, "comparisons_per_session_batch": per,
                      "steady_window_value": (n * per / span if n > 0 and span > 0 else None),
                      "paillier_bits": args.pbits, "batches_per_session": args.batches, "chunks": args.chunks, "fixed_base_window": args.window,
                      "seconds_per_session_batch": {"min": gaps[0], "median": gaps[len(gaps) // 2], "max": gaps[-1]},
                      "wire_bytes_per_comparison_sent_by_the_initiator": wire.STATS["bytes"] / (per * len(done)),
                      "rows_decrypting_to_x_le_y": sum(v["ok"] for _, v in outs), "rows_checked": sum(v["rows"] for _, v in outs),
                      "transport": "Unix socket per session, communicator.StreamCommunicator (sendall from / recv_into pinned message buffers), wire.py byte messages; two OS processes, one GPU, own HIP contexts"}), flush=True)


def parent(args):
    with tempfile.TemporaryDirectory() as td:
        sock = os.path.join(td, "sc.sock")
        common = [sys.executable, os.path.abspath(__file__), "--socket", sock, "--batch", str(args.batch), "--l", str(args.l), "--pbits", str(args.pbits),
                  "--batches", str(args.batches), "--chunks", str(args.chunks), "--window", str(args.window), "--sessions", str(args.sessions)]
        bob = subprocess.Popen(common + ["--role", "keyholder"])            # children started fresh, by a parent that has not touched the GPU
        alice = subprocess.Popen(common + ["--role", "initiator"])
        try:
            rc_a = alice.wait(timeout=1100)
            rc_b = bob.wait(timeout=120)
        finally:
            for child in (alice, bob):                  # never leave a player behind on the GPU (ended by PID)
                if child.poll() is None:
                    child.kill()
        if rc_a or rc_b:
            raise SystemExit(f"two-process run failed: initiator {rc_a}, keyholder {rc_b}")


if __name__ == "__main__":
    a = parse()
    if a.role == "parent":
        parent(a)
    elif a.role == "keyholder":
        _in_threads(a, lambda i, eng, prepared: keyholder(a, i, eng, prepared))
    else:
        import threading

        _start = (threading.Barrier(a.sessions), [])
        initiator_report(a, _in_threads(a, lambda i, eng, prepared: initiator(a, i, eng, prepared)))
