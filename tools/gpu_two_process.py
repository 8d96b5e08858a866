"""Alice and Bob as TWO OS PROCESSES on one GPU (round-4 review, "what's missing" 4): the reference's players are separate processes or
hosts (SC/test/integration/test_pool.py:41-73, test/conftest.py:162-198); round 4 measured the byte transport between two threads of
one process only.

    python tools/gpu_two_process.py [--batch 65536] [--l 32] [--pbits 2048] [--batches 6] [--chunks 1] [--window 20]

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
    return ap.parse_args()


def _keys(args):
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "keys.json")))
    pj, dj = k[f"paillier_{args.pbits}"], k[f"dgk_{args.pbits if args.pbits < 3072 else 2048}_l{args.l}"]
    H = lambda d, name: int(d[name], 16)  # noqa: E731
    return (H(pj, "p"), H(pj, "q")), {n: (H(dj, n) if n != "t" else dj["t"]) for n in ("p", "q", "g", "h", "u", "t", "v_p", "v_q")}


async def keyholder(args):
    import torch

    from protocols.secure_comparison_amd import DGK, KeyHolder, Paillier, StreamCommunicator, wire
    from protocols.secure_comparison_amd.schemes import default_engine

    (p, q), d = _keys(args)
    eng = default_engine()
    bob_p = Paillier(p * q, p, q, engine=eng)
    bob_d = DGK(d["p"] * d["q"], d["g"], d["h"], d["u"], d["t"], d["p"], d["q"], d["v_p"], d["v_q"], engine=eng, randomizer_bits=400,
                fixed_base_window=args.window)
    bob_d.prepare()
    _ = bob_p.key
    comm = await StreamCommunicator.accept_unix(args.socket, alloc=wire.pinned_buffer)       # batch messages land in pinned host memory
    bob = KeyHolder(args.l, comm, "initiator", bob_p, bob_d)
    for _ in range(args.batches + 1):                      # one warm-up batch, then the timed ones
        await bob.perform_secure_comparison_batch()
    res, expect = wire.unpack_many(await comm.recv("initiator", "check"), eng.device, expect=2)
    dec = bob_p.decrypt_raw_batch(res.contiguous())
    ok = int(((dec[:, 0] == expect.reshape(-1).to(torch.int32)) & (dec[:, 1:] == 0).all(dim=1)).sum().item())
    await comm.send("initiator", json.dumps({"ok": ok, "rows": int(res.shape[0])}).encode(), "verdict")
    await comm.close()


async def initiator(args):
    import torch

    import bench
    from protocols.secure_comparison_amd import DGK, Initiator, Paillier, StreamCommunicator, wire
    from protocols.secure_comparison_amd.schemes import default_engine

    (p, q), d = _keys(args)
    eng = default_engine()
    pub_p = Paillier(p * q, engine=eng)                         # public parts only, as received keys would be
    pub_d = DGK(d["p"] * d["q"], d["g"], d["h"], d["u"], d["t"], engine=eng, randomizer_bits=400, fixed_base_window=args.window)
    x, y, x_enc, y_enc, _ = bench.synth_inputs(eng, args.l, pub_p, pub_p, pub_d, args.batch, 400, seed=0)
    expect = (x <= y).to(torch.int32)
    comm = await StreamCommunicator.open_unix(args.socket, alloc=wire.pinned_buffer, wait_s=300)
    alice = Initiator(args.l, comm, "keyholder")
    await alice.perform_secure_comparison_batch(x_enc, y_enc, engine=eng, chunks=args.chunks)         # warm-up: tables, programs, pinned pools
    torch.cuda.synchronize()
    wire.reset_stats()
    stamps = [time.perf_counter()]
    res = None
    for _ in range(args.batches):
        res = await alice.perform_secure_comparison_batch(x_enc, y_enc, engine=eng, chunks=args.chunks)
        torch.cuda.synchronize()
        stamps.append(time.perf_counter())
    await comm.send("keyholder", wire.pack_many(res.contiguous(), expect), "check")
    verdict = json.loads(bytes(await comm.recv("keyholder", "verdict")).decode())
    await comm.close()
    per = sorted(b - a for a, b in zip(stamps, stamps[1:]))
    total = stamps[-1] - stamps[0]
    print(json.dumps({"two_process": True, "value": args.batch * args.batches / total, "unit": "comparisons/s", "batch": args.batch, "l": args.l,
                      "paillier_bits": args.pbits, "batches": args.batches, "chunks": args.chunks, "fixed_base_window": args.window,
                      "seconds_per_batch": {"min": per[0], "median": per[len(per) // 2], "max": per[-1]},
                      "wire_bytes_per_comparison_sent_by_the_initiator": wire.STATS["bytes"] / (args.batch * args.batches),
                      "rows_decrypting_to_x_le_y": verdict["ok"], "rows_checked": verdict["rows"],
                      "transport": "Unix socket, communicator.StreamCommunicator (sendall from / recv_into pinned message buffers), wire.py byte messages; two OS processes, one GPU, own HIP contexts"}), flush=True)


def parent(args):
    with tempfile.TemporaryDirectory() as td:
        sock = os.path.join(td, "sc.sock")
        common = [sys.executable, os.path.abspath(__file__), "--socket", sock, "--batch", str(args.batch), "--l", str(args.l), "--pbits", str(args.pbits),
                  "--batches", str(args.batches), "--chunks", str(args.chunks), "--window", str(args.window)]
        bob = subprocess.Popen(common + ["--role", "keyholder"])            # children started fresh, by a parent that has not touched the GPU
        alice = subprocess.Popen(common + ["--role", "initiator"])
        rc_a = alice.wait(timeout=1100)
        rc_b = bob.wait(timeout=120)
        if rc_a or rc_b:
            raise SystemExit(f"two-process run failed: initiator {rc_a}, keyholder {rc_b}")


if __name__ == "__main__":
    a = parse()
    if a.role == "parent":
        parent(a)
    else:
        asyncio.run(keyholder(a) if a.role == "keyholder" else initiator(a))
