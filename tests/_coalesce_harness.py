"""Shared by the CPU and the GPU test of the session coalescer: N perform_secure_comparison sessions on ONE Initiator / KeyHolder
pair, every `secrets` draw of a session taken from that session's own seeded stream (a context variable names the session: asyncio
tasks keep their context), every message recorded -- concurrently through the coalescer, or one session after the other without it."""
import asyncio
import contextvars
import random
import secrets
import warnings

from _comm import DictionaryCommunicator

from protocols.secure_comparison_amd import Initiator, KeyHolder

_stream: contextvars.ContextVar = contextvars.ContextVar("session_stream")


def run_sessions(pairs, l, bob_p, bob_d, coalesce: bool, alice_paillier=None, strict: bool = True, seed: int = 5, replay: bool = True):
    """(results as integers, {msg_id: [ciphertext integers]}, coalescer statistics of both players)."""
    sent = {}

    class Tap(DictionaryCommunicator):
        async def send(self, party_id, message, msg_id):
            if not msg_id.startswith("schemes"):
                flat = []
                for m in (message if isinstance(message, tuple) else (message,)):
                    flat += [c.peek_value() for c in (m if isinstance(m, list) else [m])]
                sent[msg_id] = flat
            await super().send(party_id, message, msg_id)

    box = {}
    alice = Initiator(l, Tap(box), "bob", alice_paillier)
    bob = KeyHolder(l, Tap(box), "alice", bob_p, bob_d)
    alice.coalesce_sessions = bob.coalesce_sessions = coalesce
    if replay:      # every draw through `secrets`, one call per value, in the order the single path makes them: the seeded streams replay
        from protocols.secure_comparison_amd.host_draws import SecretsDraws

        alice.draw_source = bob.draw_source = SecretsDraws()

    async def as_session(tag, i, coro_fn):
        _stream.set(random.Random(f"{seed}:{tag}:{i}"))
        return await coro_fn()

    async def together():
        a = [asyncio.ensure_future(as_session("a", i, lambda x=x, y=y: alice.perform_secure_comparison(x, y))) for i, (x, y) in enumerate(pairs)]
        b = [asyncio.ensure_future(as_session("b", i, bob.perform_secure_comparison)) for i in range(len(pairs))]
        out = await asyncio.gather(*a)
        await asyncio.gather(*b)
        return out

    async def one_by_one():
        out = []
        for i, (x, y) in enumerate(pairs):
            a = asyncio.ensure_future(as_session("a", i, lambda x=x, y=y: alice.perform_secure_comparison(x, y)))
            b = asyncio.ensure_future(as_session("b", i, bob.perform_secure_comparison))
            out.append(await a)
            await b
        return out

    real = secrets.randbelow, secrets.randbits
    secrets.randbelow, secrets.randbits = (lambda n: _stream.get().randrange(n)), (lambda k: _stream.get().getrandbits(k))
    try:
        with warnings.catch_warnings():
            if strict:  # the reference's strict fixtures: randomness / ciphertext warnings are errors (test/conftest.py:26-36)
                warnings.filterwarnings("error", ".*ciphertext", UserWarning)
                warnings.filterwarnings("error", ".*randomness", UserWarning)
            res = asyncio.run(together() if coalesce else one_by_one())
    finally:
        secrets.randbelow, secrets.randbits = real
    stats = {"alice": dict(alice._coalescer().stats), "bob": dict(bob._coalescer().stats)}
    return [r.peek_value() for r in res], sent, stats
