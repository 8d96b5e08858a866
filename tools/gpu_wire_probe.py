"""Dev tool: where a byte-transport batch of the interactive protocol spends its wall-clock time, per session thread.

    python tools/gpu_wire_probe.py [sessions] [chunks] [stagger_ms]

Wraps the batch steps and the wire functions with timers (host wall clock; a step's time includes the synchronisations it
contains) and prints one session's timeline of the last batch plus per-phase totals, then the throughput of the run."""
import asyncio
import json
import os
import sys
import threading
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from protocols.secure_comparison_amd import DGK, Initiator, KeyHolder, Paillier, wire  # noqa: E402
from protocols.secure_comparison_amd.communicator import InMemoryCommunicator  # noqa: E402
from protocols.secure_comparison_amd.engine import Engine  # noqa: E402

sessions = int(sys.argv[1]) if len(sys.argv) > 1 else 2
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 1
stagger = float(sys.argv[3]) / 1e3 if len(sys.argv) > 3 else 0.0
B, l, rbits = 65536, 32, 400
LOG = []
T0 = [0.0]


def timed(name, fn):
    def wrap(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            LOG.append((threading.get_ident(), name, t - T0[0], time.perf_counter() - T0[0]))
    return wrap


for cls, names in ((Initiator, ("step_1_batch", "step_4_batch", "step_6_7_batch")), (KeyHolder, ("step_2_4b_batch", "step_4j_5_batch"))):
    for n in names:
        setattr(cls, n, staticmethod(timed(n, getattr(cls, n))))
wire.unpack_many = timed("unpack", wire.unpack_many)
_orig_pack = wire._pack


def _pack(tensors, framed, asynchronous=False):
    t = time.perf_counter()
    m = _orig_pack(tensors, framed, asynchronous)
    LOG.append((threading.get_ident(), "pack-issue", t - T0[0], time.perf_counter() - T0[0]))
    if asynchronous:
        orig_finish = m._finish

        def fin():
            LOG.append((threading.get_ident(), "pack-done", t - T0[0], time.perf_counter() - T0[0]))
            return orig_finish()
        m._finish = fin
    return m


wire._pack = _pack
_orig_finish = wire.Reserved.finish


async def _finish(self):
    t = time.perf_counter()
    try:
        return await _orig_finish(self)
    finally:
        LOG.append((threading.get_ident(), "message written by the step's launches (wait)", t - T0[0], time.perf_counter() - T0[0]))


wire.Reserved.finish = _finish

keys = json.load(open(bench.KEYS))
pj, dj = keys["paillier_2048"], keys["dgk_2048_l32"]
p, q = int(pj["p"], 16), int(pj["q"], 16)
H = lambda k: int(dj[k], 16)  # noqa: E731
engines = [Engine() for _ in range(sessions)]
parts = []
for i, e in enumerate(engines):
    bob_p = Paillier(p * q, p, q, engine=e)
    bob_d = DGK(H("p") * H("q"), H("g"), H("h"), H("u"), dj["t"], H("p"), H("q"), H("v_p"), H("v_q"), engine=e, randomizer_bits=rbits, fixed_base_window=16)
    alice_d = bob_d.public_copy()
    if i:
        bob_d.share_tables_from(parts[0][3])
        alice_d.share_tables_from(parts[0][1])
    alice_d.prepare(), bob_d.prepare()
    parts.append((bob_p.public_copy(), alice_d, bob_p, bob_d, torch.cuda.Stream()))
    e.set_chip_share(sessions)
x, y, x_enc, y_enc, _ = bench.synth_inputs(engines[0], l, parts[0][0], parts[0][2], parts[0][3], B, rbits, seed=0)
per = B // sessions
inputs = [(x_enc[i * per:(i + 1) * per].contiguous(), y_enc[i * per:(i + 1) * per].contiguous()) for i in range(sessions)]
torch.cuda.synchronize()
reps = int(os.environ.get("WP_REPS", "5"))
DONE = []
results = [None] * sessions


def session(i):
    ap, ad, bp, bd, stream = parts[i]
    comm = InMemoryCommunicator(device_tensors=False)
    alice, bob = Initiator(l, comm, "k", ap, ad), KeyHolder(l, comm.peer(), "i", bp, bd)
    time.sleep(stagger * i)
    with torch.cuda.stream(stream):
        for _ in range(reps):
            async def go():
                r, _ = await asyncio.gather(alice.perform_secure_comparison_batch(*inputs[i], engine=engines[i], chunks=chunks), bob.perform_secure_comparison_batch())
                return r
            results[i] = asyncio.run(go())
            stream.synchronize()
            DONE.append((time.perf_counter() - T0[0], i))



for warm in (True, False):
    LOG.clear()
    DONE.clear()
    T0[0] = time.perf_counter()
    ths = [threading.Thread(target=session, args=(i,)) for i in range(sessions)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    torch.cuda.synchronize()
    dt = time.perf_counter() - T0[0]
print(f"sessions {sessions} chunks {chunks} stagger {stagger * 1e3:.0f} ms: {reps} batches per session in {dt * 1e3:.1f} ms = {B * reps / dt:.0f} comparisons/s "
      f"({dt / reps * 1e3:.1f} ms per batch of {B})")
done = sorted(DONE)
print("batch completions (ms, session):", [(round(t * 1e3), i) for t, i in done])
if len(done) > sessions:
    n, span = len(done) - sessions, done[-1][0] - done[sessions - 1][0]
    print(f"steady window: {n} session-batches of {per} in {span * 1e3:.1f} ms = {n * per / span:.0f} comparisons/s")
tids = sorted({t for t, *_ in LOG})
for tid in tids[:2]:
    ev = [e for e in LOG if e[0] == tid]
    last = ev[len(ev) * (reps - 1) // reps:]
    print(f"-- session thread {tids.index(tid)}: last batch")
    for _, name, a, b in last:
        print(f"   {a * 1e3:9.1f} .. {b * 1e3:9.1f}  ({(b - a) * 1e3:7.1f} ms)  {name}")
tot = {}
for _, name, a, b in LOG:
    tot[name] = tot.get(name, 0.0) + (b - a)
print({k: round(v * 1e3 / reps / sessions, 1) for k, v in tot.items()}, "ms per batch and session")
