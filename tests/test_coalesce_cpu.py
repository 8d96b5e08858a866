"""Concurrent single comparisons coalesced into batch calls (coalesce.StepCoalescer), CPU tier: the host logic on the test-only
engine.  The GPU twin of the main test is tests/test_gpu_round5.py::test_concurrent_sessions_are_coalesced_into_batch_launches."""
import asyncio
import os
import sys
import warnings

import pytest

sys.path.insert(0, os.path.dirname(__file__))
from _coalesce_harness import run_sessions  # noqa: E402
from _comm import DictionaryCommunicator  # noqa: E402
from _oracle_engine import OracleEngine  # noqa: E402
from conftest import oracle_dgk, oracle_paillier  # noqa: E402

from protocols.secure_comparison_amd import DGK, Initiator, KeyHolder, Paillier  # noqa: E402
from protocols.secure_comparison_amd.coalesce import StepCoalescer  # noqa: E402

L = 16


@pytest.fixture(scope="module")
def world(keys):
    osk, od = oracle_paillier(keys, 1024), oracle_dgk(keys, "dgk_tiny_l16")
    eng = OracleEngine()
    bob_p = Paillier(osk.n, osk.p, osk.q, engine=eng)
    bob_d = DGK(od.n, od.g, od.h, od.u, od.t, od.p, od.q, od.v_p, od.v_q, full_decryption=True, engine=eng, randomizer_bits=50)
    return eng, bob_p, bob_d


def test_concurrent_sessions_equal_the_uncoalesced_runs_message_by_message(world):
    """24 concurrent perform_secure_comparison sessions on ONE Initiator / KeyHolder pair (the reference's test_parallel_runs shape,
    test/unit/test_secure_comparison.py:804-835, scaled up), strict warnings: every message of every session equals what the same
    session sends when it runs alone and uncoalesced with the same random stream; the steps really ran as batches."""
    from protocols.secure_comparison_amd import coalesce

    eng, bob_p, bob_d = world
    pairs = [(23, 42), (42, 23), (7, 7), (-3, 5), (5, -3), (0, 0), (65535, 1), (1, 65535)] * 3
    before = dict(coalesce.STACK_STATS)
    co_res, co_sent, stats = run_sessions(pairs, L, bob_p, bob_d, coalesce=True)
    # both players coalesced the same sessions in the same order: the three big per-session arrays ([d] | [beta_i], [c_i], the three
    # Paillier ciphertexts) went from one batched call to the next as the peer's own array, not block by block
    assert coalesce.STACK_STATS["taken_whole"] - before["taken_whole"] == 3 and coalesce.STACK_STATS["assembled"] == before["assembled"]
    un_res, un_sent, _ = run_sessions(pairs, L, bob_p, bob_d, coalesce=False)
    assert co_sent.keys() == un_sent.keys() and len(co_sent) == 4 * len(pairs)
    for k in co_sent:
        assert co_sent[k] == un_sent[k], k
    assert co_res == un_res
    assert [bob_p.decrypt(bob_p._ct_class(v, bob_p)) for v in co_res] == [int(x <= y) for x, y in pairs]
    for side in ("alice", "bob"):
        assert stats[side]["largest"] == len(pairs) and stats[side]["fallbacks"] == 0
    assert stats["alice"]["calls"] == 3 and stats["bob"]["calls"] == 3          # (the stand-in engine has no background context: 2 + the no-op)
    assert not bob_p._pool and not bob_d._pool


def test_ciphertext_inputs_and_mixed_inputs(world):
    """Sessions may bring ciphertexts or plaintexts (SC/initiator.py:69-72, :93-102), in any mix, inside one batch."""
    eng, bob_p, bob_d = world
    pub = bob_p.public_copy()
    vals = [(3, 9), (9, 3), (4, 4), (-8, -9)]
    pairs = [(pub.unsafe_encrypt(x), pub.unsafe_encrypt(y)) if i % 2 else ((pub.unsafe_encrypt(x), y) if i == 0 else (x, y)) for i, (x, y) in enumerate(vals)]
    res, _, stats = run_sessions(pairs, L, bob_p, bob_d, coalesce=True, alice_paillier=pub)
    assert [bob_p.decrypt(bob_p._ct_class(v, bob_p)) for v in res] == [int(x <= y) for x, y in vals]
    assert stats["alice"]["largest"] == 4


def test_one_bad_session_fails_alone(world):
    """A session whose input makes the batched call fail (here: a ciphertext that is not invertible modulo N^2) gets its own
    exception; its neighbours in the batch complete."""
    eng, bob_p, bob_d = world
    pub = bob_p.public_copy()
    bad = pub._ct_class(pub.public_key.n, pub)                    # N is not a unit modulo N^2
    box = {}
    alice, bob = Initiator(L, DictionaryCommunicator(box), "bob", pub, bob_d.public_copy()), KeyHolder(L, DictionaryCommunicator(box), "alice", bob_p, bob_d)

    async def go():
        tasks = [asyncio.ensure_future(alice.perform_secure_comparison(x, y)) for x, y in ((1, 2), (bad, 5), (9, 2))]
        bobs = [asyncio.ensure_future(bob.perform_secure_comparison()) for _ in range(3)]
        done = await asyncio.gather(*tasks, return_exceptions=True)
        for t in bobs:                                            # the key holder's session of the failed comparison waits for a message that never comes
            if not t.done():
                t.cancel()
        await asyncio.gather(*bobs, return_exceptions=True)
        return done

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = asyncio.run(go())
    assert isinstance(out[1], (ZeroDivisionError, ValueError))     # the product: NotInvertibleError; the stand-in engine: what pow raises
    assert bob_p.decrypt(out[0]) == 1 and bob_p.decrypt(out[2]) == 0
    assert alice._coalescer().stats["fallbacks"] == 1


def test_coalescer_scheduling():
    """Requests of one turn of the event loop share a call; max_batch cuts a queue; linger waits for stragglers; a result count that
    does not match is an error for every waiter; cancellation of one waiter leaves the others alone."""
    calls = []

    def run(items):
        calls.append(list(items))
        return [10 * i for i in items]

    async def burst(co, n):
        return await asyncio.gather(*(co.submit("k", run, i) for i in range(n)))

    co = StepCoalescer()
    assert asyncio.run(burst(co, 5)) == [0, 10, 20, 30, 40] and calls == [[0, 1, 2, 3, 4]]
    calls.clear()
    co = StepCoalescer(max_batch=2)
    assert asyncio.run(burst(co, 5)) == [0, 10, 20, 30, 40] and [len(c) for c in calls] == [2, 2, 1]
    calls.clear()

    async def straggler(co):
        async def late():
            await asyncio.sleep(0.02)
            return await co.submit("k", run, 7)
        return await asyncio.gather(co.submit("k", run, 1), late())

    assert asyncio.run(straggler(StepCoalescer(linger_s=0.2))) == [10, 70] and calls == [[1, 7]]
    calls.clear()
    assert asyncio.run(straggler(StepCoalescer())) == [10, 70] and calls == [[1], [7]]

    async def short(co):
        return await asyncio.gather(co.submit("k", lambda items: [1], 1), co.submit("k", lambda items: [1], 2), return_exceptions=True)

    out = asyncio.run(short(StepCoalescer()))
    assert all(isinstance(o, (RuntimeError, int)) for o in out)        # the batch of two fails, the single re-runs return one result each
    calls.clear()

    async def cancelled(co):
        a = asyncio.ensure_future(co.submit("k", run, 1))
        b = asyncio.ensure_future(co.submit("k", run, 2))
        await asyncio.sleep(0)
        a.cancel()
        return await asyncio.gather(a, b, return_exceptions=True)

    out = asyncio.run(cancelled(StepCoalescer(linger_s=0.05)))
    assert isinstance(out[0], asyncio.CancelledError) and out[1] == 20


def test_a_step_runs_the_moment_every_session_in_flight_waits_in_it():
    """A player's sessions register with its coalescer (StepCoalescer.session): a queue that holds a request of EVERY session in flight
    cannot grow, so it is executed without the quiet turn of the event loop a partial queue waits for -- a lone session pays no extra
    turn, a burst in lock step goes when its last session arrives; a session's `first` request never takes that short cut (sessions
    created together may not all have started), and results come back in the order the sessions queued."""
    calls, turns = [], []

    def run(items):
        calls.append(list(items))
        return [10 * i for i in items]

    async def go():
        co = StepCoalescer()
        loop = asyncio.get_running_loop()

        async def session(i):
            with co.session():
                a = await co.submit("one", run, i, first=True)
                t0 = loop.time()
                b = await co.submit("two", run, i + 100)
                turns.append(loop.time() - t0)
                return a, b

        out = await asyncio.gather(*(session(i) for i in range(5)))
        assert co.active == 0
        lone = await session(9)
        return out, lone

    out, lone = asyncio.run(go())
    assert out == [(10 * i, 10 * (i + 100)) for i in range(5)] and lone == (90, 1090)
    assert calls == [[0, 1, 2, 3, 4], [100, 101, 102, 103, 104], [9], [109]]


def test_the_linger_restarts_after_every_straggler():
    """Requests that trickle in over a transport -- each less than linger_s after the one before, the whole burst longer than linger_s --
    still share ONE call: the wait for stragglers starts again after every arrival (it used to run once per queue: the second
    straggler found a queue that had lingered already and was executed without the third); a gap longer than linger_s cuts."""
    calls = []

    def run(items):
        calls.append(list(items))
        return items

    async def trickle(co, gaps):
        async def one(i, at):
            await asyncio.sleep(at)
            return await co.submit("k", run, i)

        at, jobs = 0.0, []
        for i, g in enumerate(gaps):
            at += g
            jobs.append(one(i, at))
        return await asyncio.gather(*jobs)

    assert asyncio.run(trickle(StepCoalescer(linger_s=0.08), [0, 0.03, 0.03, 0.03, 0.03, 0.03])) == list(range(6))
    assert calls == [[0, 1, 2, 3, 4, 5]]
    calls.clear()
    asyncio.run(trickle(StepCoalescer(linger_s=0.05), [0, 0.01, 0.01, 0.3, 0.01]))
    assert calls == [[0, 1, 2], [3, 4]]


def test_a_forked_child_does_not_repeat_its_parents_draws():
    """HostDraws buffers a megabyte of the generator's stream and pools values ahead: after a fork the child's copy of all that is
    dropped (os.register_at_fork), so parent and child never hand out the same randomizer."""
    import multiprocessing as mp

    from protocols.secure_comparison_amd.host_draws import HostDraws

    d = HostDraws()
    d.randbelow(1 << 200), d.bits_rows(100, 3), d.below_rows_nonzero((1 << 40) + 15, 3), d.permutation(9)     # buffers and pools are warm

    def child(q):
        q.put((d.randbelow(1 << 200), d.bits_rows(100, 2).tolist(), d.below_rows_nonzero((1 << 40) + 15, 2).tolist(), d.permutation(9)))

    ctx = mp.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=child, args=(q,))
    p.start()
    theirs = q.get(timeout=60)
    p.join(30)
    mine = (d.randbelow(1 << 200), d.bits_rows(100, 2).tolist(), d.below_rows_nonzero((1 << 40) + 15, 2).tolist(), d.permutation(9))
    assert theirs[0] != mine[0] and theirs[1] != mine[1] and theirs[2] != mine[2]


def test_the_collector_pauses_for_the_start_of_a_burst_and_no_longer():
    """coalesce._CollectorPause: off when the first session of a burst enters, on again when the last one leaves -- or limit_s after
    the first entered, and then not again before every session has left; never touched when the application had it off or when
    pause_collector_s is 0; two coalescers of one process share the one state."""
    import gc
    import time

    seen = {}

    async def burst(cos, hold_s=0.0, probe=None):
        async def session(co, i):
            with co.session():
                seen.setdefault("inside", []).append(gc.isenabled())
                await co.submit("k", lambda items: items, i)
                if hold_s:
                    await asyncio.sleep(hold_s)
                    await co.submit("k2", lambda items: items, i)       # (the check runs after a batched call)
                    await asyncio.sleep(0)
                    seen.setdefault("late", []).append(gc.isenabled())
        await asyncio.gather(*(session(co, i) for i in range(4) for co in cos))

    assert gc.isenabled()
    try:
        asyncio.run(burst([StepCoalescer(), StepCoalescer()]))
        assert seen.pop("inside") == [False] * 8 and gc.isenabled()
        asyncio.run(burst([StepCoalescer(pause_collector_s=0.05)], hold_s=0.12))
        assert seen.pop("inside") == [False] * 4 and seen.pop("late") == [True] * 4 and gc.isenabled()
        asyncio.run(burst([StepCoalescer(pause_collector_s=0)]))
        assert seen.pop("inside") == [True] * 4 and gc.isenabled()
        gc.disable()                                                   # the application's own choice stays
        asyncio.run(burst([StepCoalescer()]))
        assert seen.pop("inside") == [False] * 4 and not gc.isenabled()
    finally:
        gc.enable()
