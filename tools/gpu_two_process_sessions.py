"""N concurrent SINGLE comparisons between two OS processes on one GPU: the reference's own call shape (one `perform_secure_comparison`
per comparison, concurrency by asyncio sessions: SC/initiator.py:69-175, :86-87; test/unit/test_secure_comparison.py:804-835) with the
players where the reference has them -- in separate processes (SC/test/integration/test_pool.py:41-73).

    python tools/gpu_two_process_sessions.py [--sessions 1024] [--bursts 5] [--l 32] [--pbits 2048] [--linger-ms 2] [--quiet-gc 0]

The parent never touches the GPU; the key holder and the initiator are fresh children with their own HIP contexts, library contexts and
generators, a Unix socket between them (communicator.StreamCommunicator: the scheme pair as its public document, every ciphertext
message in wire.pack_session_message's form).  Each process coalesces its own sessions' steps into batch launches (coalesce.py).  The
initiator's inputs are plaintext integers (encrypted inside her step-1 launch).  After the timed bursts she ships her last results
and the expected bits to the key holder, who decrypts them and answers with the number of correct rows.  Prints one JSON object."""
import argparse
import asyncio
import json
import os
import random
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
    ap.add_argument("--sessions", type=int, default=1024)
    ap.add_argument("--bursts", type=int, default=5)
    ap.add_argument("--l", type=int, default=32)
    ap.add_argument("--pbits", type=int, default=2048)
    ap.add_argument("--linger-ms", type=float, default=2.0)
    ap.add_argument("--quiet-gc", type=int, default=0)
    ap.add_argument("--pause-gc", type=float, default=0.5)         # the players' coalesce_pause_collector_s (0: collector untouched)
    ap.add_argument("--profile", default="")
    return ap.parse_args()


def _keys(args):
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "keys.json")))
    pj, dj = k[f"paillier_{args.pbits}"], k[f"dgk_{args.pbits}_l{args.l}"]
    H = lambda d, name: int(d[name], 16)  # noqa: E731
    return (H(pj, "p"), H(pj, "q")), {n: (H(dj, n) if n != "t" else dj["t"]) for n in ("p", "q", "g", "h", "u", "t", "v_p", "v_q")}


def _collector(args):
    import contextlib

    from protocols.secure_comparison_amd.coalesce import quiet_collector

    return quiet_collector() if args.quiet_gc else contextlib.nullcontext()


async def keyholder(args):
    import torch

    from protocols.secure_comparison_amd import DGK, KeyHolder, Paillier, StreamCommunicator, wire
    from protocols.secure_comparison_amd.schemes import default_engine

    (p, q), d = _keys(args)
    eng = default_engine()
    bob_p = Paillier(p * q, p, q, engine=eng)
    bob_d = DGK(d["p"] * d["q"], d["g"], d["h"], d["u"], d["t"], d["p"], d["q"], d["v_p"], d["v_q"], engine=eng, randomizer_bits=400)
    bob_d.prepare()
    _ = bob_p.key
    comm = await StreamCommunicator.accept_unix(args.socket, engine=eng)
    bob = KeyHolder(args.l, comm, "initiator", bob_p, bob_d)
    bob.coalesce_linger_s = args.linger_ms / 1e3
    bob.coalesce_pause_collector_s = args.pause_gc
    with _collector(args):
        for _ in range(args.bursts + 1):                      # one warm-up burst, then the timed ones
            await asyncio.gather(*(bob.perform_secure_comparison() for _ in range(args.sessions)))
    res, expect = wire.unpack_many(await comm.recv("initiator", "check"), eng.device, expect=2)
    dec = bob_p.decrypt_raw_batch(res.contiguous())
    ok = int(((dec[:, 0] == expect.reshape(-1).to(torch.int32)) & (dec[:, 1:] == 0).all(dim=1)).sum().item())
    st = bob._coalescer().stats
    await comm.send("initiator", json.dumps({"ok": ok, "rows": int(res.shape[0]), "calls": st["calls"], "items": st["items"], "largest": st["largest"],
                                             "fallbacks": st["fallbacks"], "seconds": st["seconds"]}).encode(), "verdict")
    await comm.close()


async def initiator(args):
    import torch

    from protocols.secure_comparison_amd import Initiator, StreamCommunicator, wire
    from protocols.secure_comparison_amd.coalesce import rows_of
    from protocols.secure_comparison_amd.schemes import default_engine

    eng = default_engine()
    comm = await StreamCommunicator.open_unix(args.socket, engine=eng, wait_s=300)
    alice = Initiator(args.l, comm, "keyholder")               # her schemes arrive over the wire
    alice.coalesce_linger_s = args.linger_ms / 1e3
    alice.coalesce_pause_collector_s = args.pause_gc
    rng = random.Random(11)
    xs = [rng.randrange(1 << args.l) for _ in range(args.sessions)]
    ys = [xs[i] if i % 4 == 0 else rng.randrange(1 << args.l) for i in range(args.sessions)]

    async def burst():
        return await asyncio.gather(*(alice.perform_secure_comparison(x, y) for x, y in zip(xs, ys)))

    spans, res = [], None
    with _collector(args):
        await burst()                                          # warm-up: key objects, tables, programs, the first scheme document
        wire.reset_stats()
        prof = None
        if args.profile:
            import cProfile

            prof = cProfile.Profile()
            prof.enable()
        for _ in range(args.bursts):
            t0 = time.perf_counter()
            res = await burst()
            spans.append(time.perf_counter() - t0)
        if prof is not None:
            import pstats

            prof.disable()
            with open(args.profile, "w") as fh:
                pstats.Stats(prof, stream=fh).sort_stats("cumulative").print_stats(45)
                pstats.Stats(prof, stream=fh).sort_stats("tottime").print_stats(35)
    sent = wire.STATS["bytes"]
    pai = alice.scheme_paillier
    rows = eng.upload_words(rows_of(list(res), 2 * pai.mod_n.nwords))
    expect = torch.tensor([int(x <= y) for x, y in zip(xs, ys)], dtype=torch.int32, device=eng.device)
    await comm.send("keyholder", wire.pack_many(rows.contiguous(), expect), "check")
    verdict = json.loads(bytes(await comm.recv("keyholder", "verdict")).decode())
    st = alice._coalescer().stats
    await comm.close()
    spans.sort()
    print(json.dumps({"two_process_sessions": True, "value": args.sessions * args.bursts / sum(spans), "unit": "comparisons/s",
                      "best_burst_value": args.sessions / spans[0], "sessions": args.sessions, "bursts": args.bursts, "l": args.l,
                      "paillier_bits": args.pbits, "linger_ms": args.linger_ms, "quiet_collector": bool(args.quiet_gc), "pause_collector_s": args.pause_gc,
                      "seconds_per_burst": {"min": spans[0], "median": spans[len(spans) // 2], "max": spans[-1]},
                      "bytes_sent_by_the_initiator_per_comparison": sent / (args.sessions * args.bursts),
                      "initiator_batched_calls": {"calls": st["calls"], "items": st["items"], "largest": st["largest"], "fallbacks": st["fallbacks"],
                                                  "seconds_inside_all_bursts": st["seconds"]},
                      "keyholder_batched_calls": {k: verdict[k] for k in ("calls", "items", "largest", "fallbacks")} | {"seconds_inside_all_bursts": verdict["seconds"]},
                      "rows_decrypting_to_x_le_y": verdict["ok"], "rows_checked": verdict["rows"],
                      "transport": "Unix socket, communicator.StreamCommunicator; ciphertext messages as wire.pack_session_message bytes; two OS processes, one GPU"}),
          flush=True)


def parent(args):
    with tempfile.TemporaryDirectory() as td:
        sock = os.path.join(td, "sc.sock")
        common = [sys.executable, os.path.abspath(__file__), "--socket", sock, "--sessions", str(args.sessions), "--bursts", str(args.bursts),
                  "--l", str(args.l), "--pbits", str(args.pbits), "--linger-ms", str(args.linger_ms), "--quiet-gc", str(args.quiet_gc), "--pause-gc", str(args.pause_gc)]
        prof = ["--profile", args.profile] if args.profile else []
        bob = subprocess.Popen(common + ["--role", "keyholder"])            # children started fresh, by a parent that has not touched the GPU
        alice = subprocess.Popen(common + prof + ["--role", "initiator"])
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
    else:
        asyncio.run(keyholder(a) if a.role == "keyholder" else initiator(a))
