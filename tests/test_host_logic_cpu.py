"""Host-side logic of the product (step choreography, operator algebra, bit-major batch layout, protocol driver,
error behaviour) exercised on the CPU with the test-only OracleEngine standing in for the HIP engine.
The arithmetic itself is checked on the GPU (tests/test_gpu_*.py)."""
import asyncio
import random
import warnings

import pytest
import torch

from _comm import DictionaryCommunicator
from _oracle_engine import OracleEngine
from conftest import oracle_dgk, oracle_paillier
from oracle import sc_oracle as o
from protocols.secure_comparison_amd import DGK, Initiator, KeyHolder, Paillier
from protocols.secure_comparison_amd.batch import BatchDraws, BatchTrace, secure_comparison_batch

L = 16


@pytest.fixture(scope="module")
def world(keys):
    osk, od = oracle_paillier(keys, 1024), oracle_dgk(keys, "dgk_tiny_l16")
    eng = OracleEngine()
    bob_p = Paillier(osk.n, osk.p, osk.q, engine=eng)
    bob_d = DGK(od.n, od.g, od.h, od.u, od.t, od.p, od.q, od.v_p, od.v_q, full_decryption=True, engine=eng, randomizer_bits=50)
    return osk, od, eng, bob_p, bob_d


def test_operator_algebra_matches_oracle(world):
    osk, od, eng, pai, dgk = world
    a, b = pai.unsafe_encrypt(5), pai.unsafe_encrypt(-3)
    assert (a + b).value == osk.add(osk.enc_raw(5), osk.enc_raw(osk.encode(-3)))
    assert (a - b).value == osk.add(osk.enc_raw(5), osk.neg(osk.enc_raw(osk.encode(-3))))
    assert (1 - a).value == osk.add(osk.neg(osk.enc_raw(5)), osk.enc_raw(1))
    assert (a * 7).value == osk.mul(osk.enc_raw(5), 7) and (a * -1).value == osk.neg(osk.enc_raw(5))
    assert pai.decrypt(a - b) == 8 and pai.decrypt(b) == -3 and pai.decrypt(b, apply_encoding=False) == osk.n - 3
    c = dgk.unsafe_encrypt(-1, apply_encoding=False)
    assert c.value == od.enc_raw(-1) == o.mod_inv(od.g, od.n)           # g^-1, not g^(u-1)  (SURVEY 8(a) note 2)
    assert (c * 0).value == 1 and (3 * c).value == od.mul(c.value, 3) and (2 + c).value == od.add(c.value, od.enc_raw(2))
    assert dgk.decrypt(dgk.unsafe_encrypt(5) + dgk.unsafe_encrypt(9)) == 14 and dgk.is_zero(dgk.unsafe_encrypt(0))


def test_single_steps_match_oracle(world):
    osk, od, eng, pai, dgk = world
    rng = random.Random(4)
    for x, y in ((23, 42), (42, 23), (-400, -383), (7, 7)):
        dr = o.draw(rng, L, osk, od, 50, shuffle=True)
        x_enc, y_enc = pai.unsafe_encrypt(x), pai.unsafe_encrypt(y)
        z_enc, r = Initiator.step_1(x_enc, y_enc, L, pai, r=dr.r)
        z, beta = KeyHolder.step_2(z_enc, L, pai)
        alpha = Initiator.step_3(r, L)
        d_enc = KeyHolder.step_4a(z, dgk, pai, L)
        beta_enc = KeyHolder.step_4b(beta, L, dgk)
        d_enc = Initiator.step_4c(d_enc, r, dgk, pai)
        xor = Initiator.step_4d(alpha, beta_enc)
        w, alpha_tilde = Initiator.step_4e(r, alpha, xor, d_enc, pai)
        w = Initiator.step_4f(w)
        s, delta_a = Initiator.step_4g(dr.delta_a)
        c = Initiator.step_4h(s, alpha, alpha_tilde, d_enc, beta_enc, w, delta_a, dgk)
        c = Initiator.step_4i(c, dgk, do_shuffle=True, rhos=dr.rhos, permutation=dr.perm)
        delta_b = KeyHolder.step_4j(c, dgk)
        z1, z2, db = KeyHolder.step_5(z, L, delta_b, pai)
        res = Initiator.step_7(z1, z2, r, L, Initiator.step_6(delta_a, db), pai)
        assert res.value == o.compare(x_enc.value, y_enc.value, L, osk, od, dr, randomize=False)
        assert pai.decrypt(res) == int(x <= y)


def test_step_guards(world):
    osk, od, eng, pai, dgk = world
    with pytest.raises(AssertionError):
        Initiator.step_1(pai.unsafe_encrypt(1), pai.unsafe_encrypt(2), osk.n.bit_length() - 2, pai)
    for r in (-10, osk.n + 10):
        with pytest.raises(AssertionError):
            Initiator.step_4c(dgk.unsafe_encrypt(0), r, dgk, pai)
    with pytest.raises(AssertionError):
        KeyHolder.step_4a(5, dgk, pai, 40)  # u must exceed 2^(l+2)
    with pytest.raises(ValueError):
        Initiator(L).scheme_paillier
    with pytest.raises(ValueError):
        KeyHolder(L).scheme_dgk
    with pytest.raises(ValueError):
        asyncio.run(Initiator(L).perform_secure_comparison(1, 2))
    with pytest.raises(ValueError):
        asyncio.run(KeyHolder(L).perform_secure_comparison())


def _interactive(world, x, y, strict=True, sessions=1):
    osk, od, eng, bob_p, bob_d = world
    box = {}
    alice = Initiator(L, communicator=DictionaryCommunicator(box), other_party="bob")
    bob = KeyHolder(L, communicator=DictionaryCommunicator(box), other_party="alice", scheme_paillier=bob_p, scheme_dgk=bob_d)

    async def go():
        out = []
        for _ in range(sessions):
            res, _ = await asyncio.gather(alice.perform_secure_comparison(x, y), bob.perform_secure_comparison())
            out.append(res)
        return out

    with warnings.catch_warnings():
        if strict:  # the reference's strict fixtures: randomness / ciphertext warnings are errors (test/conftest.py:26-36)
            warnings.filterwarnings("error", ".*ciphertext", UserWarning)
            warnings.filterwarnings("error", ".*randomness", UserWarning)
        return asyncio.run(go()), alice, bob


@pytest.mark.parametrize("x, y", [(23, 42), (42, 23), (-1, 0), (5, 5)])
def test_interactive_protocol_exact_randomness_budget(world, x, y):
    res, alice, bob = _interactive(world, x, y, strict=True, sessions=2)
    assert [world[3].decrypt(r) for r in res] == [int(x <= y)] * 2
    assert alice.session_id == 2 and bob.session_id == 2
    # the boot counts 1/(l+1) and 3/(l+1) are exactly what one comparison consumes: pools are empty afterwards
    assert not alice.scheme_paillier._pool and not alice.scheme_dgk._pool and not world[3]._pool and not world[4]._pool


def _run_single(world, x, y, fused, seed):
    """One interactive comparison with every `secrets` draw of both players replaced by a seeded stream; all wire values."""
    import secrets

    osk, od, eng, bob_p, bob_d = world
    rng = random.Random(seed)
    real = secrets.randbelow, secrets.randbits
    secrets.randbelow, secrets.randbits = (lambda n: rng.randrange(n)), (lambda k: rng.getrandbits(k))
    sent = {}

    class Tap(DictionaryCommunicator):
        async def send(self, party_id, message, msg_id):
            flat = []
            for m in (message if isinstance(message, tuple) else (message,)):
                flat += [c.peek_value() for c in (m if isinstance(m, list) else [m]) if hasattr(c, "peek_value")]
            sent[msg_id] = flat
            await super().send(party_id, message, msg_id)

    try:
        box = {}
        alice, bob = Initiator(L, Tap(box), "bob"), KeyHolder(L, Tap(box), "alice", bob_p, bob_d)
        # fused: True = the default path (steps as batch launches through the session coalescer; its draws replayed through `secrets`),
        # "alone" = the same five step-level calls per session without the coalescer (round 4's path), False = one launch per operator
        alice.fuse_steps = bob.fuse_steps = bool(fused)
        alice.coalesce_sessions = bob.coalesce_sessions = fused is True
        from protocols.secure_comparison_amd.host_draws import SecretsDraws

        alice.draw_source = bob.draw_source = SecretsDraws()

        async def go():
            res, _ = await asyncio.gather(alice.perform_secure_comparison(x, y), bob.perform_secure_comparison())
            return res

        res = asyncio.run(go())
    finally:
        secrets.randbelow, secrets.randbits = real
    assert not bob_p._pool and not bob_d._pool and not alice.scheme_dgk._pool and not alice.scheme_paillier._pool
    return res.peek_value(), sent


@pytest.mark.parametrize("x, y", [(23, 42), (42, 23), (-5, -5)])
def test_fused_single_comparison_equals_the_operator_path(world, x, y):
    """perform_secure_comparison through the step-level library calls on one-element batches (the default) sends and returns
    the same ciphertexts, bit for bit, as the reference-shaped body that walks the ciphertext operator algebra step by step --
    with the same random stream: every message of the exchange is compared, not only the result."""
    fused, sent_f = _run_single(world, x, y, True, 99)
    alone, sent_a = _run_single(world, x, y, "alone", 99)
    plain, sent_p = _run_single(world, x, y, False, 99)
    assert sent_f.keys() == sent_p.keys() == sent_a.keys() and all(sent_f[k] == sent_p[k] == sent_a[k] for k in sent_f if not k.startswith("schemes"))
    assert fused == plain == alone and world[3].decrypt(world[3]._ct_class(fused, world[3])) == int(x <= y)
    assert _run_single(world, x, y, True, 100)[0] != fused          # another stream, other ciphertexts


def test_parallel_sessions_are_namespaced(world):
    osk, od, eng, bob_p, bob_d = world
    box = {}
    a1 = Initiator(L, DictionaryCommunicator(box), "bob", session_id=0)
    a2 = Initiator(L, DictionaryCommunicator(box), "bob", session_id=10)
    b1 = KeyHolder(L, DictionaryCommunicator(box), "alice", bob_p, bob_d, session_id=0)
    b2 = KeyHolder(L, DictionaryCommunicator(box), "alice", bob_p, bob_d, session_id=10)

    async def go():
        return await asyncio.gather(a1.perform_secure_comparison(3, 9), a2.perform_secure_comparison(9, 3),
                                    b1.perform_secure_comparison(), b2.perform_secure_comparison())

    r = asyncio.run(go())
    assert bob_p.decrypt(r[0]) == 1 and bob_p.decrypt(r[1]) == 0


def test_mismatching_scheme_raises_valueerror(world, keys):
    osk, od, eng, bob_p, bob_d = world
    other = oracle_paillier(keys, 2048)
    box = {}
    alice = Initiator(L, DictionaryCommunicator(box), "bob", scheme_paillier=Paillier(other.n, engine=eng))
    bob = KeyHolder(L, DictionaryCommunicator(box), "alice", bob_p, bob_d)

    async def go():
        await asyncio.gather(alice.perform_secure_comparison(1, 2), bob.make_and_send_encryption_schemes(1))

    with pytest.raises(ValueError, match=".*Paillier"):
        asyncio.run(go())


def test_randomize_warnings(world):
    osk, od, eng, pai, dgk = world
    pai.shut_down()
    ct = pai.unsafe_encrypt(3)
    with pytest.warns(UserWarning, match=".*randomness"):
        ct.randomize()                       # pool empty -> generated on the fly
    with pytest.warns(UserWarning, match=".*ciphertext"):
        pai.boot_randomness_generation(1)
        ct.randomize()                       # already fresh
    assert pai.decrypt(ct) == 3


def make_draws(eng, drs, l, nw, ew, er):
    B = len(drs)
    bm = lambda rows, w: torch.stack([eng.upload([rows[b][i] for b in range(B)], w) for i in range(l + 1)])  # noqa: E731
    inv_rc = [[None] * (l + 1) for _ in range(B)]
    for b, d in enumerate(drs):   # the oracle randomizes after the shuffle: map r_c back to pre-shuffle positions
        for k, src in enumerate(d.perm):
            inv_rc[b][src] = d.r_c[k]
    return BatchDraws(r=eng.upload([d.r for d in drs], nw), delta_a=eng.upload_u64([d.delta_a for d in drs]),
                      rhos=bm([d.rhos for d in drs], ew), permutation=torch.tensor([d.perm for d in drs], dtype=torch.int64),
                      rho_z=eng.upload([d.rho_z for d in drs], nw), r_bob_dgk=bm([[d.r_d] + d.r_beta for d in drs], er),
                      r_alice_dgk=bm(inv_rc, er), rho_zeta_1=eng.upload([d.rho_zeta1 for d in drs], nw),
                      rho_zeta_2=eng.upload([d.rho_zeta2 for d in drs], nw), rho_delta_b=eng.upload([d.rho_delta_b for d in drs], nw))


@pytest.mark.parametrize("use_crt", [False, True])
def test_batch_driver_layout_and_shuffle(world, use_crt):
    """Bit-major layouts, the permutation gather, fused randomizer ordering and the CRT recombination, vs the oracle."""
    osk, od, eng, _, bob_d = world
    bob_p = Paillier(osk.n, osk.p, osk.q, engine=eng, use_crt=use_crt)
    rng = random.Random(77)
    B = 7  # ragged on purpose
    xs = [rng.randrange(1 << L) for _ in range(B)]
    ys = [xs[i] if i % 3 == 0 else rng.randrange(1 << L) for i in range(B)]
    drs = [o.draw(rng, L, osk, od, 50) for _ in range(B)]
    x_enc = [osk.randomize(osk.enc_raw(x), 1 + rng.randrange(osk.n - 1)) for x in xs]
    y_enc = [osk.randomize(osk.enc_raw(y), 1 + rng.randrange(osk.n - 1)) for y in ys]
    traces = [dict() for _ in range(B)]
    expect = [o.compare(a, b, L, osk, od, d, True, t) for a, b, d, t in zip(x_enc, y_enc, drs, traces)]
    nw = bob_p.mod_n.nwords
    draws = make_draws(eng, drs, L, nw, (od.u.bit_length() + 31) // 32, 2)
    tr = BatchTrace()
    got = secure_comparison_batch(eng.upload(x_enc, 2 * nw), eng.upload(y_enc, 2 * nw), L, bob_p.public_copy(), bob_d.public_copy(),
                                  bob_p, bob_d, draws, True, tr)
    assert eng.download(got) == expect
    # the same batch with the randomizer exponentiations taken out of the steps and computed on a second context (`side`)
    from protocols.secure_comparison_amd.batch import PartySet

    eng2 = OracleEngine()
    bob_p2 = Paillier(osk.n, osk.p, osk.q, engine=eng2, use_crt=use_crt)
    bob_d2 = DGK(od.n, od.g, od.h, od.u, od.t, od.p, od.q, od.v_p, od.v_q, engine=eng2, randomizer_bits=50)
    side = PartySet(bob_p2.public_copy(), bob_d2.public_copy(), bob_p2, bob_d2, None)
    tr2 = BatchTrace()
    got2 = secure_comparison_batch(eng.upload(x_enc, 2 * nw), eng.upload(y_enc, 2 * nw), L, bob_p.public_copy(), bob_d.public_copy(),
                                   bob_p, bob_d, draws, True, tr2, side=side)
    assert eng.download(got2) == expect and torch.equal(tr2.c_sent, tr.c_sent) and torch.equal(tr2.beta_enc, tr.beta_enc)
    assert [eng.download(tr.c_sent[:, b])for b in range(B)] == [t["c_enc"] for t in traces]
    assert tr.delta_b.tolist() == [t["delta_b"] for t in traces]
    assert [osk.dec_raw(v) for v in eng.download(got)] == [int(x <= y) for x, y in zip(xs, ys)]


@pytest.mark.parametrize("device_tensors", [True, False])
def test_batched_interactive_protocol_and_wire_format(world, device_tensors):
    """perform_secure_comparison_batch on both sides over the dictionary transport, with the arrays handed over as they are
    (device_tensors) and serialized into one byte buffer per message: (1) with injected draws it is bit-identical to the
    oracle; (2) with draws=None every random input comes from the engine's generator (here: its CPU restatement) -- results
    decrypt correctly, two runs differ, and the keyed stream makes a run reproducible."""
    from protocols.secure_comparison_amd import wire

    osk, od, eng, bob_p, bob_d = world
    rng = random.Random(31)
    B = 5
    xs = [rng.randrange(1 << L) for _ in range(B)]
    ys = [xs[i] if i % 2 == 0 else rng.randrange(1 << L) for i in range(B)]
    x_enc = [osk.randomize(osk.enc_raw(x), 1 + rng.randrange(osk.n - 1)) for x in xs]
    y_enc = [osk.randomize(osk.enc_raw(y), 1 + rng.randrange(osk.n - 1)) for y in ys]
    nw = bob_p.mod_n.nwords
    tx, ty = eng.upload(x_enc, 2 * nw), eng.upload(y_enc, 2 * nw)
    drs = [o.draw(rng, L, osk, od, 50) for _ in range(B)]
    draws = make_draws(eng, drs, L, nw, (od.u.bit_length() + 31) // 32, 2)
    seen = []
    for use_draws, key in ((True, None), (False, bytes(range(32))), (False, bytes(range(32))), (False, bytes(32))):
        box = {}
        alice = Initiator(L, DictionaryCommunicator(box, device_tensors), "bob")
        bob = KeyHolder(L, DictionaryCommunicator(box, device_tensors), "alice", bob_p, bob_d)
        if key is not None:
            eng.rng_seed(key)
        wire.reset_stats()

        async def go():
            res, _ = await asyncio.gather(alice.perform_secure_comparison_batch(tx, ty, draws if use_draws else None, engine=eng),
                                          bob.perform_secure_comparison_batch(draws if use_draws else None))
            return res

        with warnings.catch_warnings():
            warnings.filterwarnings("error", ".*randomness", UserWarning)
            got = eng.download(asyncio.run(go()))
        assert [osk.dec_raw(v) for v in got] == [int(x <= y) for x, y in zip(xs, ys)]
        if use_draws:
            assert got == [o.compare(a, b, L, osk, od, d, True) for a, b, d in zip(x_enc, y_enc, drs)]
        else:
            seen.append(got)
        assert (wire.STATS["device_arrays"] == 7 and wire.STATS["bytes"] == 0) if device_tensors else \
            (wire.STATS["device_arrays"] == 0 and wire.STATS["bytes"] > B * (2 * (L + 1) * 4 * bob_d.mod_n.nwords))
        assert alice.scheme_paillier == bob_p and alice.scheme_dgk == bob_d and not box
    assert seen[0] == seen[1] and seen[0] != seen[2]        # same key, same stream; another key, other ciphertexts


@pytest.mark.parametrize("device_tensors", [True, False])
def test_chunked_batch_equals_the_unchunked_one(world, device_tensors):
    """A batch sent as k sub-sessions on one connection (Initiator.perform_secure_comparison_batch(chunks=k): plan message, message
    ids `.._chunk_i`, the chunks' steps interleaving on the event loop while messages are packed asynchronously) gives the same
    ciphertexts bit for bit as the single session with the same injected draws -- ragged chunk sizes, more chunks than
    comparisons, and the key holder learning the plan from the first message alone."""
    osk, od, eng, bob_p, bob_d = world
    rng = random.Random(77)
    B = 7
    xs = [rng.randrange(1 << L) for _ in range(B)]
    ys = [xs[i] if i % 3 == 0 else rng.randrange(1 << L) for i in range(B)]
    x_enc = [osk.randomize(osk.enc_raw(x), 1 + rng.randrange(osk.n - 1)) for x in xs]
    y_enc = [osk.randomize(osk.enc_raw(y), 1 + rng.randrange(osk.n - 1)) for y in ys]
    nw = bob_p.mod_n.nwords
    tx, ty = eng.upload(x_enc, 2 * nw), eng.upload(y_enc, 2 * nw)
    drs = [o.draw(rng, L, osk, od, 50) for _ in range(B)]
    draws = make_draws(eng, drs, L, nw, (od.u.bit_length() + 31) // 32, 2)
    expect = [o.compare(a, b, L, osk, od, d, True) for a, b, d in zip(x_enc, y_enc, drs)]
    for chunks, use_draws in ((1, True), (3, True), (7, True), (50, True), (2, False)):
        box = {}
        alice = Initiator(L, DictionaryCommunicator(box, device_tensors), "bob")
        bob = KeyHolder(L, DictionaryCommunicator(box, device_tensors), "alice", bob_p, bob_d)

        async def go():
            res, _ = await asyncio.gather(alice.perform_secure_comparison_batch(tx, ty, draws if use_draws else None, engine=eng, chunks=chunks),
                                          bob.perform_secure_comparison_batch(draws if use_draws else None))
            return res

        got = eng.download(asyncio.run(go()))
        assert [osk.dec_raw(v) for v in got] == [int(x <= y) for x, y in zip(xs, ys)]
        if use_draws:
            assert got == expect
        assert not box
    # a plan that does not match the key holder's injected draws, and a chunk of another size than announced, are refused
    from protocols.secure_comparison_amd import wire

    with pytest.raises(ValueError):
        wire.plan_of(wire.PLAN_MAGIC + b'{"chunks": [0]}')
    with pytest.raises(ValueError):
        wire.plan_of(wire.PLAN_MAGIC + b'not json')
    assert wire.plan_of(wire.pack_many(tx)) is None and wire.plan_of(wire.pack_plan([4, 3])) == [4, 3]


def test_bad_permutation_is_refused_before_the_send(world):
    """An injected row that is not a permutation is caught on Alice's side before [c_i] leaves (the scatter target is
    zero-filled, never stale memory)."""
    osk, od, eng, bob_p, bob_d = world
    rng = random.Random(5)
    B = 3
    nw = bob_p.mod_n.nwords
    x_enc = [osk.enc_raw(rng.randrange(1 << L)) for _ in range(B)]
    drs = [o.draw(rng, L, osk, od, 50) for _ in range(B)]
    draws = make_draws(eng, drs, L, nw, (od.u.bit_length() + 31) // 32, 2)
    draws.permutation[1, 0] = draws.permutation[1, 1]
    assert not bool(Initiator.permutation_is_valid(draws.permutation))
    box = {}
    alice = Initiator(L, DictionaryCommunicator(box), "bob")
    bob = KeyHolder(L, DictionaryCommunicator(box), "alice", bob_p, bob_d)

    async def go():
        t = eng.upload(x_enc, 2 * nw)
        b_task = asyncio.ensure_future(bob.perform_secure_comparison_batch(draws))
        try:
            await alice.perform_secure_comparison_batch(t, t, draws, engine=eng)
        finally:
            b_task.cancel()

    with pytest.raises(ValueError, match="permutation"):
        asyncio.run(go())
    assert "step_4i_batch_session_1" not in box


def test_wire_and_randomness_helpers():
    from protocols.secure_comparison_amd import randomness as R
    from protocols.secure_comparison_amd import wire
    from protocols.secure_comparison_amd.limbs import words_to_ints
    import numpy as np

    n = (1 << 1023) + 99
    eng = OracleEngine()
    g = torch.Generator().manual_seed(5)
    for source, gen in (("device", None), ("torch", g)):
        t = R.uniform_below(n, 300, eng, source, gen, nonzero=True)
        vals = words_to_ints(t.numpy().view(np.uint32))
        assert all(0 < v < n for v in vals) and len(set(vals)) == 300
        bits = R.random_bits(35, (4, 7), eng, source, gen)
        assert bits.shape == (4, 7, 2) and int((bits[..., 1].to(torch.int64) & 0xFFFFFFFF).max()) < 8
        perms = R.random_permutations(6, 17, eng, source, gen)
        assert perms.dtype == torch.int64 and all(sorted(p.tolist()) == list(range(17)) for p in perms)
        coins = R.random_coins(700, eng, source, gen)
        assert coins.dtype == torch.int64 and set(coins.tolist()) == {0, 1}
    with pytest.raises(ValueError):
        R.uniform_below(n, 3, "cpu")        # the device source needs the engine whose generator draws
    a, b, c = wire.unpack_many(wire.pack_many(t, perms, torch.tensor([1, 0], dtype=torch.uint8)))
    assert torch.equal(a, t) and torch.equal(b, perms) and c.dtype == torch.uint8
    assert torch.equal(wire.unpack_tensor(bytes(wire.pack_tensor(perms))), perms)      # immutable bytes from a socket
    assert wire.unpack_tensor(wire.pack_tensor(torch.zeros((0, 4), dtype=torch.int32))).shape == (0, 4)
    with pytest.raises(ValueError):
        wire.unpack_tensor(b"nope" + bytes(16))
    with pytest.raises(ValueError):
        wire.incoming(wire.DeviceArrays((t,)), "cpu", expect=2)
    with pytest.raises(ValueError):
        wire.incoming(wire.DeviceArrays(("not an array",)), "cpu", expect=1)


def test_freshness_discipline(world):
    """The fresh-flag rules the reference's strict fixtures rely on (test/conftest.py:26-36, README.md:152-154)."""
    osk, od, eng, pai, dgk = world
    pai.shut_down()
    pai.boot_randomness_generation(3)
    a = pai.unsafe_encrypt(4)
    assert not a.fresh
    a.randomize()
    assert a.fresh
    with pytest.warns(UserWarning, match=".*ciphertext"):
        b = a + 1                      # a fresh ciphertext consumed by a homomorphic operation
    assert not a.fresh and not b.fresh and pai.decrypt(b) == 5
    comm = DictionaryCommunicator({})
    with pytest.warns(UserWarning, match=".*ciphertext"):
        asyncio.run(comm.send("bob", b, msg_id="m1"))   # non-fresh on the wire: randomized first, with a warning
    got = asyncio.run(comm.recv("alice", msg_id="m1"))
    assert not got.fresh and got.peek_value() != osk.enc_raw(5) and pai.decrypt(got) == 5
    c = pai.unsafe_encrypt(9)
    c.randomize()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        asyncio.run(comm.send("bob", c, msg_id="m2"))    # fresh ciphertext: no warning
    assert not c.fresh                 # disclosed
    assert c.peek_value() == asyncio.run(comm.recv("alice", msg_id="m2")).peek_value()
    pai.shut_down()


def test_pooled_batch_mode(world):
    """randomize="pool": randomizers come from pre-booted device pools; results decrypt correctly and the pools are
    consumed exactly."""
    from protocols.secure_comparison_amd.batch import boot_pools

    osk, od, eng, bob_p, bob_d = world
    alice_p, alice_d = bob_p.public_copy(), bob_d.public_copy()
    for sch in (bob_p, bob_d):
        sch.shut_down()
    rng = random.Random(3)
    B = 4
    xs = [rng.randrange(1 << L) for _ in range(B)]
    ys = [rng.randrange(1 << L) for _ in range(B)]
    drs = [o.draw(rng, L, osk, od, 50) for _ in range(B)]
    nw = bob_p.mod_n.nwords
    draws = make_draws(eng, drs, L, nw, (od.u.bit_length() + 31) // 32, 2)
    boot_pools(B, L, alice_p, alice_d, bob_p, bob_d)
    with warnings.catch_warnings():
        warnings.filterwarnings("error", ".*randomness", UserWarning)
        res = secure_comparison_batch(eng.upload([osk.enc_raw(x) for x in xs], 2 * nw), eng.upload([osk.enc_raw(y) for y in ys], 2 * nw),
                                      L, alice_p, alice_d, bob_p, bob_d, draws, randomize="pool")
    assert [osk.dec_raw(v) for v in eng.download(res)] == [int(x <= y) for x, y in zip(xs, ys)]
    assert all(s._batch_pool.shape[0] == 0 for s in (alice_p, alice_d, bob_p, bob_d))


def test_in_memory_transport_hands_over_public_schemes_only(world):
    """What Alice receives over the in-memory transport is what a serializing transport would give her: the PUBLIC parts of Bob's
    schemes (the reference's serializers drop the secret key) and ciphertexts bound to them -- never Bob's own scheme objects."""
    from protocols.secure_comparison_amd.communicator import InMemoryCommunicator

    _, _, _, bob_p, bob_d = world
    comm = InMemoryCommunicator()

    async def go():
        bob_p.boot_randomness_generation(1)
        ct = bob_p.unsafe_encrypt(5)
        ct.randomize()
        await comm.send("alice", (bob_p, bob_d, [ct]), msg_id="m")
        return await comm.peer().recv("bob", msg_id="m")

    got_p, got_d, (got_ct,) = asyncio.run(go())
    assert got_p is not bob_p and got_p.secret_key is None and got_p == bob_p
    assert got_d is not bob_d and got_d.secret_key is None and got_d == bob_d
    assert got_ct.scheme is got_p and got_ct.peek_value() > 0
    assert bob_p.for_wire() is got_p and got_p.for_wire() is got_p          # one public copy per scheme object


def test_background_randomness_falls_back_to_blocking_without_a_gpu_engine(world):
    """boot_randomness_generation(background=True) on an engine that has no second context to offer (the CPU tier's stand-in)
    generates at once: nothing pending, the pool filled, the values rho^N mod N^2."""
    osk, _, _, bob_p, _ = world
    bob_p.shut_down()
    bob_p.boot_randomness_generation(3, background=True)
    assert not bob_p._pending and len(bob_p._pool) == 3
    n2 = osk.n * osk.n
    for v in list(bob_p._pool):
        assert 0 < v < n2 and pow(v, (osk.p - 1) * (osk.q - 1), n2) == 1      # an N-th power: its order divides lambda
    bob_p.shut_down()
    assert not bob_p._pool
