"""GPU tests added in round 4: the step-4i shuffle taken from the permutation inside the blinding store (rows that are not
permutations act as the identity), the fused step-1 / step-6/7 programs and the single comparison through the step-level calls
against the operator path, the chunked byte transport against the single session, shard results written into one array, the
measured policy constants and the in-kernel clock probe."""
import asyncio
import random

import pytest
import torch

from conftest import oracle_dgk, oracle_paillier
from test_gpu_parity import _schemes

pytestmark = pytest.mark.gpu


def test_shuffle_from_the_permutation_inside_the_store(engine, keys):
    """Initiator.step_4i_batch with a batch of permutations equals blinding without one followed by a gather -- for l + 1 = 17, 33
    and 65 planes (the latter crosses the 64-bit word of the validity mask), with the fixed-base tail, on batches that end in a
    partially filled wave; a row that repeats an index, or names one out of range (negative, l + 1, 2^40), acts as the identity
    for that comparison and for no other."""
    from protocols.secure_comparison_amd import DGK, Initiator

    rng = random.Random(11)
    for name, l, B in (("dgk_1024_l16", 16, 37), ("dgk_2048_l32", 32, 50), ("dgk_2048_l64", 64, 21)):
        dgk = oracle_dgk(keys, name)
        sch = DGK(dgk.n, dgk.g, dgk.h, dgk.u, dgk.t, engine=engine, randomizer_bits=400, fixed_base_window=8)
        nw, ew = sch.mod_n.nwords, (dgk.u.bit_length() + 31) // 32
        c = engine.upload([rng.randrange(1, dgk.n) for _ in range((l + 1) * B)], nw).reshape(l + 1, B, nw)
        rhos = engine.upload([rng.randrange(1, dgk.u) for _ in range((l + 1) * B)], ew).reshape(l + 1, B, ew)
        r = engine.upload([rng.getrandbits(400) for _ in range((l + 1) * B)], 13).reshape(l + 1, B, 13)
        pm = torch.stack([torch.randperm(l + 1) for _ in range(B)]).to(engine.device)
        for rr in (None, r):
            plain = Initiator.step_4i_batch(c, sch, rhos, None, rr)
            shuffled = Initiator.step_4i_batch(c, sch, rhos, pm, rr)
            idx = pm.t().reshape(l + 1, B, 1).expand(l + 1, B, nw)
            assert torch.equal(shuffled, torch.gather(plain, 0, idx)), (name, rr is None)
        bad = pm.clone()
        bad[3, 0] = bad[3, 1]                      # a repeated index
        bad[7, l] = l + 1                          # out of range by one
        bad[11, 2] = -1
        bad[B - 1, 5] = 1 << 40
        got = Initiator.step_4i_batch(c, sch, rhos, bad, None)
        plain = Initiator.step_4i_batch(c, sch, rhos, None, None)
        eff = pm.clone()
        for b in (3, 7, 11, B - 1):
            eff[b] = torch.arange(l + 1, device=engine.device)
        idx = eff.t().reshape(l + 1, B, 1).expand(l + 1, B, nw)
        assert torch.equal(got, torch.gather(plain, 0, idx)), name
        assert not bool(Initiator.permutation_is_valid(bad)) and bool(Initiator.permutation_is_valid(pm))


def _run_single(engine, bob_p, bob_d, l, x, y, fused, seed):
    import secrets

    from _comm import DictionaryCommunicator
    from protocols.secure_comparison_amd import Initiator, KeyHolder

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
        alice, bob = Initiator(l, Tap(box), "bob"), KeyHolder(l, Tap(box), "alice", bob_p, bob_d)
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
    return res, sent


@pytest.mark.parametrize("pbits, dname, l", [(1024, "dgk_1024_l16", 16), (2048, "dgk_2048_l32", 32)])
def test_fused_single_comparison_equals_the_operator_path(engine, keys, pbits, dname, l):
    """BASELINE configs[0] (x = 23, y = 42, l = 16, 1024-bit keys) and a 2048-bit / l = 32 comparison: perform_secure_comparison
    through the five step-level library calls on one-element batches sends and returns the same ciphertexts, bit for bit, as the
    reference-shaped body that launches one kernel per ciphertext operator -- every message of the exchange compared, under the
    same random stream; and the result decrypts to x <= y."""
    from protocols.secure_comparison_amd import DGK, Paillier

    sk, dgk = oracle_paillier(keys, pbits), oracle_dgk(keys, dname)
    bob_p = Paillier(sk.n, sk.p, sk.q, engine=engine)
    bob_d = DGK(dgk.n, dgk.g, dgk.h, dgk.u, dgk.t, dgk.p, dgk.q, dgk.v_p, dgk.v_q, engine=engine, randomizer_bits=400)
    for x, y in ((23, 42), (42, 23), (-7, -7)):
        fused, sent_f = _run_single(engine, bob_p, bob_d, l, x, y, True, 5)
        alone, sent_a = _run_single(engine, bob_p, bob_d, l, x, y, "alone", 5)
        plain, sent_p = _run_single(engine, bob_p, bob_d, l, x, y, False, 5)
        assert sent_f.keys() == sent_p.keys() == sent_a.keys()
        for k in sent_f:
            assert sent_f[k] == sent_p[k] == sent_a[k], k
        assert fused.peek_value() == plain.peek_value() == alone.peek_value() and bob_p.decrypt(fused) == int(x <= y)


def test_chunked_byte_transport_equals_the_single_session(engine, keys):
    """The interactive batch protocol over the byte transport with the batch cut into chunks (plan message, sub-sessions whose
    messages are packed on the copy stream while the next chunk computes) returns the same ciphertexts as the single session with
    the same injected draws, for both wire forms; with draws=None every row still decrypts to x <= y."""
    import bench
    from protocols.secure_comparison_amd import Initiator, KeyHolder
    from protocols.secure_comparison_amd.communicator import InMemoryCommunicator

    sk, dgk = oracle_paillier(keys, 1024), oracle_dgk(keys, "dgk_1024_l16")
    l, B = 16, 3001
    alice_p, alice_d, bob_p, bob_d = _schemes(engine, sk, dgk, 400)
    x, y, x_enc, y_enc, draws = bench.synth_inputs(engine, l, alice_p, bob_p, bob_d, B, 400, seed=9, shuffle=True)
    expect = (x <= y).to(torch.int32)
    results = {}
    for device_tensors in (True, False):
        for chunks, use_draws in ((1, True), (3, True), (4, False)):
            comm = InMemoryCommunicator(device_tensors=device_tensors)
            alice, bob = Initiator(l, comm, "bob", alice_p, alice_d), KeyHolder(l, comm.peer(), "alice", bob_p, bob_d)

            async def go():
                res, _ = await asyncio.gather(alice.perform_secure_comparison_batch(x_enc, y_enc, draws if use_draws else None, engine=engine, chunks=chunks),
                                              bob.perform_secure_comparison_batch(draws if use_draws else None))
                return res

            res = asyncio.run(go())
            dec = bob_p.decrypt_raw_batch(res)
            assert bool(((dec[:, 0] == expect) & (dec[:, 1:] == 0).all(dim=1)).all().item()), (device_tensors, chunks)
            if use_draws:
                results[(device_tensors, chunks)] = res
    ref = results[(True, 1)]
    assert all(torch.equal(ref, r) for r in results.values())


def test_shard_results_land_in_one_array(engine, keys):
    """ConcurrentShards.run(out=...) writes every shard's rows into its block of one preallocated array: equal to the
    concatenation of the per-shard results and to the single-stream batch."""
    import bench
    from protocols.secure_comparison_amd import DGK, Paillier
    from protocols.secure_comparison_amd.batch import ConcurrentShards, PartySet, secure_comparison_batch, split_draws
    from protocols.secure_comparison_amd.distributed import shard_bounds
    from protocols.secure_comparison_amd.engine import Engine

    sk, dgk = oracle_paillier(keys, 1024), oracle_dgk(keys, "dgk_1024_l16")
    l, B = 16, 777
    alice_p, alice_d, bob_p, bob_d = _schemes(engine, sk, dgk, 400)
    x, y, x_enc, y_enc, draws = bench.synth_inputs(engine, l, alice_p, bob_p, bob_d, B, 400, seed=3, shuffle=True)
    whole = secure_comparison_batch(x_enc, y_enc, l, alice_p, alice_d, bob_p, bob_d, draws)
    engines = [engine, Engine(), Engine()]
    sets = []
    for e in engines:
        bp = Paillier(sk.n, sk.p, sk.q, engine=e)
        bd = DGK(dgk.n, dgk.g, dgk.h, dgk.u, dgk.t, dgk.p, dgk.q, dgk.v_p, dgk.v_q, engine=e, randomizer_bits=400)
        sets.append(PartySet(bp.public_copy(), bd.public_copy(), bp, bd, torch.cuda.Stream()))
    bounds = [shard_bounds(B, i, 3) for i in range(3)]
    shards = [(x_enc[a:b].contiguous(), y_enc[a:b].contiguous(), d) for (a, b), d in zip(bounds, split_draws(draws, bounds))]
    runner = ConcurrentShards(sets)
    try:
        parts = runner.run(shards, l)
        out = torch.full_like(whole, 0x55)
        got = runner.run(shards, l, out=out)
        torch.cuda.synchronize()
        assert got is out and torch.equal(out, torch.cat(parts, dim=0)) and torch.equal(out, whole)
        with pytest.raises(ValueError):
            runner.run(shards, l, out=out[:-1])
    finally:
        runner.close()
        for e in engines[1:]:
            e.close()


def test_policy_constants_are_measured_and_the_clock_probe_reads_a_clock(engine, keys):
    """sc_ctx_policy: the round times behind the one-lane policy are measured on this device (positive, half rounds cheaper than
    full ones, a two-lane round cheaper than a one-lane round of twice the numbers); sc_clock_probe: the stamping twin of the
    dominant pair launch reports an engine clock between 1 and 2.6 GHz and computes what the timed kernel computes (the probe's
    launch is the library's own randomize path: its duration is that of the kernel)."""
    from protocols.secure_comparison_amd import Paillier

    pol = engine.policy()
    assert pol["simds"] >= 4 and all(v > 0 for v in pol.values())
    assert pol["one_lane_half_round_ms"] < pol["one_lane_full_round_ms"] and pol["two_lane_later_half_round_ms"] < pol["two_lane_full_round_ms"]
    assert 0.3 < pol["two_lane_full_round_ms"] / pol["one_lane_full_round_ms"] < 0.9
    sk = oracle_paillier(keys, 2048)
    alice_p = Paillier(sk.n, engine=engine)
    rng = random.Random(2)
    B = 16384                                   # enough items for the chip-filling (4,18) modulus-multiple launch
    rho = engine.upload([rng.randrange(1, sk.n) for _ in range(64)], alice_p.mod_n.nwords).repeat(B // 64, 1).contiguous()
    old = engine.lib.sc_ctx_set_latency_mode(engine.ctx, 1)
    try:
        ghz, ms = engine.clock_probe(alice_p.key, rho)
    finally:
        engine.set_latency_mode(0)
    assert 1.0 < ghz < 2.6 and ms > 1.0, (ghz, ms)


def test_inconsistent_dgk_secret_key_is_refused(engine, keys):
    """sc_dgk_key_create checks that the secret part belongs to the public one: p q = n and h^v_p = 1 (mod p), h^v_q = 1 (mod q)
    -- otherwise the key holder's CRT randomizers (exponents reduced modulo v_p, v_q) would silently differ from h^r mod n."""
    from protocols.secure_comparison_amd import DGK

    d = oracle_dgk(keys, "dgk_1024_l16")
    good = DGK(d.n, d.g, d.h, d.u, d.t, d.p, d.q, d.v_p, d.v_q, engine=engine, randomizer_bits=400)
    _ = good.key
    for bad in (dict(v_p=d.v_p + 2), dict(v_q=d.v_q + 2), dict(p=d.p + 2), dict(h=d.h * d.g % d.n)):
        kw = dict(n=d.n, g=d.g, h=d.h, u=d.u, t=d.t, p=d.p, q=d.q, v_p=d.v_p, v_q=d.v_q)
        kw.update(bad)
        sch = DGK(kw["n"], kw["g"], kw["h"], kw["u"], kw["t"], kw["p"], kw["q"], kw["v_p"], kw["v_q"], engine=engine, randomizer_bits=400)
        with pytest.raises(ValueError):
            _ = sch.key


def test_fixed_base_windows_above_twenty(engine):
    """Tables with 2^21 .. 2^24 rows per window (the bench's default is 24: 17 rows per 400-bit randomizer from an 82 GB table):
    h^r and the blinding launch's fused c^rho h^r against Python's pow, with exponents that hit the first, the last and random rows
    of every window.  Short exponents keep the tables of this test small (2 x 2^22 and 2 x 2^24 rows of 288 B: 2.4 / 9.7 GB)."""
    rng = random.Random(24)
    n = rng.getrandbits(2048) | (1 << 2047) | 1
    mod = engine.modulus(n)
    h = rng.randrange(2, n)
    for window, ebits in ((22, 44), (24, 48), (21, 30)):
        fb = engine.fixed_base(mod, h, ebits, window)
        top = (1 << ebits) - 1
        r = [0, 1, top, (1 << window) - 1, 1 << window, top ^ ((1 << window) - 1)] + [rng.getrandbits(ebits) for _ in range(26)]
        c = [rng.randrange(n) for _ in r]
        rho = [1 + rng.randrange((1 << 34) - 1) for _ in r]
        tr, tc, trho = engine.upload(r, 2), engine.upload(c, mod.nwords), engine.upload(rho, 2)
        assert engine.download(engine.fixedbase_pow(fb, tr)) == [pow(h, x, n) for x in r]
        got = engine.download(engine.modexp_var(mod, tc, trho, 34, fb, tr))
        assert got == [pow(y, e, n) * pow(h, x, n) % n for y, e, x in zip(c, rho, r)]
        del fb


def test_background_randomizer_generation(engine, keys, monkeypatch):
    """boot_randomness_generation(background=True): the key holder's Paillier randomizers queued on a second context and
    collected when first needed -- the same values, in the same order of use, as the blocking form (rho^N mod N^2 of the draws)."""
    import secrets as _secrets

    from protocols.secure_comparison_amd import Paillier
    from protocols.secure_comparison_amd import schemes as S

    sk = oracle_paillier(keys, 1024)
    drawn = []

    def fake_randbelow(n):
        v = (0x9E3779B97F4A7C15 * (len(drawn) + 1) ** 3 + 12345) % n
        drawn.append(v)
        return v

    monkeypatch.setattr(S.secrets, "randbelow", fake_randbelow)
    n2 = sk.n * sk.n
    for background in (True, False):
        drawn.clear()
        sch = Paillier(sk.n, sk.p, sk.q, engine=engine)
        sch.boot_randomness_generation(3, background=background)
        assert len(drawn) == 3                                          # drawn at boot time in both forms
        assert bool(sch._pending) == background
        got = [sch.get_randomness() for _ in range(3)]
        want = [pow(1 + v, sk.n, n2) for v in drawn[:3]]
        assert got == want[::-1]                                        # the pool is used from its end (list.pop)
        sch.boot_randomness_generation(2, background=background)
        sch.shut_down()                                                 # collects what is still in flight
        assert not sch._pending and not sch._pool
    assert _secrets is S.secrets


def test_segmented_pair_launch_and_forced_fork_give_the_same_residues(engine, keys):
    """A context that shares the chip runs Alice's single-round pair launch in segments (the pair parked in the slot's table between
    them); a context that owns it, or one in fork mode 2, runs the key holder's CRT halves side by side: the same residues as the
    plain forms, item by item (and equal to rho^N mod N^2 on sampled rows)."""
    from protocols.secure_comparison_amd import Paillier

    sk = oracle_paillier(keys, 2048)
    alice_p, bob_p = Paillier(sk.n, engine=engine), Paillier(sk.n, sk.p, sk.q, engine=engine)
    rng = random.Random(5)
    B = 24576                                    # between half a round and one round of k_pvm<4,18>: the segmented case at chip share 2
    base = [rng.randrange(1, sk.n) for _ in range(96)]
    rho = engine.upload(base, alice_p.mod_n.nwords).repeat(B // 96, 1).contiguous()
    plain = alice_p.randomizer_batch(rho)
    engine.set_chip_share(2)
    try:
        seg = alice_p.randomizer_batch(rho)
    finally:
        engine.set_chip_share(1)
    assert torch.equal(plain, seg)
    n2 = sk.n * sk.n
    assert engine.download(plain[:3]) == [pow(v, sk.n, n2) for v in base[:3]]
    small = rho[:6144].contiguous()
    outs = []
    for mode in (0, 2, 1):
        engine.set_fork_mode(mode)
        outs.append(bob_p.randomizer_batch(small))
    engine.set_fork_mode(1)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and torch.equal(outs[0], plain[:6144])
    # the key holder's y^p mod p^2 launches of a shard (more than half a round, up to two): segments with a table slot per item
    many = rho.repeat(2, 1).contiguous()                 # 49152 numbers: 1536 waves of k_pvm<2,18>
    whole = bob_p.randomizer_batch(many)
    engine.set_chip_share(2)
    try:
        in_segments = bob_p.randomizer_batch(many)
    finally:
        engine.set_chip_share(1)
    assert torch.equal(whole, in_segments) and torch.equal(whole[:B], plain)
