"""GPU tests added in round 3: the device-side generator against its CPU restatement, the exact configuration bench.py times
(two concurrent shards, shared tables, shuffle) against the oracle, the batched interactive protocol on device draws at a
BASELINE-sized shape, and the in-place refusal of sc_modinv."""
import asyncio
import os
import random
import sys

import pytest
import torch

from conftest import oracle_dgk, oracle_paillier
from oracle import chacha_rng as cr
from test_gpu_parity import _schemes
from test_gpu_round2 import _oracle_rows

pytestmark = pytest.mark.gpu

KEY = bytes((7 * i + 3) & 0xFF for i in range(32))


def _ints(engine, t):
    return engine.download(t.reshape(-1, t.shape[-1]))


def test_device_generator_equals_its_restatement(engine):
    """Every kind of draw, word for word against oracle/chacha_rng.py (itself pinned by the RFC 8439 / OpenSSL vectors in
    tests/test_rng_oracle_cpu.py), with the call counter advancing exactly one per call; ragged counts cross the 256-thread
    block and the 512-coin block boundaries."""
    engine.rng_seed(KEY)
    call = 0
    for bits, count in ((400, 300), (35, 1000), (2048, 70), (1, 5), (32, 257), (33, 64), (513, 3)):
        assert _ints(engine, engine.rng_bits(bits, count)) == cr.rng_bits(KEY, call, bits, count), (bits, count)
        call += 1
    rng = random.Random(1)
    n2048 = rng.getrandbits(2048) | (1 << 2047) | 1
    u35 = (1 << 34) + 25
    for n, count, nz in ((n2048, 200, False), (n2048, 130, True), (u35, 3000, True), (5, 600, False), (5, 600, True), (2, 50, True),
                         ((1 << 64) - 1, 300, False), ((1 << 32) + 1, 500, True)):
        got = _ints(engine, engine.rng_below(n, count, nz))
        assert got == cr.rng_below(KEY, call, n, count, nz), (n.bit_length(), count, nz)
        assert all((1 if nz else 0) <= v < n for v in got)
        call += 1
    for count in (1, 511, 512, 513, 5000):
        assert engine.rng_coins(count).tolist() == cr.rng_coins(KEY, call, count)
        call += 1
    for k, count in ((33, 500), (65, 130), (17, 70), (1, 3), (2, 100), (256, 5)):
        got = engine.rng_permutations(k, count)
        assert got.tolist() == cr.rng_permutations(KEY, call, k, count), (k, count)
        assert bool((torch.sort(got, dim=1).values == torch.arange(k, device=got.device)).all())
        call += 1
    # re-seeding restarts the stream; another key gives another stream; an unseeded / OS-seeded engine still draws
    engine.rng_seed(KEY)
    again = _ints(engine, engine.rng_bits(400, 300))
    assert again == cr.rng_bits(KEY, 0, 400, 300)
    engine.rng_seed(bytes(32))
    assert _ints(engine, engine.rng_bits(400, 300)) != again
    engine.rng_seed(None)
    a, b = _ints(engine, engine.rng_bits(256, 100)), _ints(engine, engine.rng_bits(256, 100))
    assert a != b and len(set(a)) == 100
    with pytest.raises(ValueError):
        engine.rng_seed(b"short")
    with pytest.raises(ValueError):
        engine.rng_permutations(300, 4)
    assert engine.rng_bits(64, 0).shape == (0, 2) and engine.rng_coins(0).shape == (0,)


def test_device_draws_are_uniform_enough(engine):
    """Coarse distribution checks at batch scale (a mis-masked top word or a biased rejection shows up here): means of the top
    words, coin balance, every shuffle position equally likely."""
    engine.rng_seed(None)
    n = (1 << 2047) + (1 << 2046) + 12345          # 0.75 * 2^2048: rejection really rejects
    r = engine.rng_below(n, 200000)
    top = (r[:, -1].to(torch.int64) & 0xFFFFFFFF).double()
    assert abs(float(top.mean()) / (0.75 * 2 ** 32) - 0.5) < 0.01 and float(top.max()) < 0.75 * 2 ** 32 + 1
    coins = engine.rng_coins(1 << 20)
    assert abs(float(coins.double().mean()) - 0.5) < 0.003
    perms = engine.rng_permutations(33, 1 << 17)
    counts = torch.zeros(33, 33, dtype=torch.int64, device=perms.device)
    for pos in range(33):
        counts[pos] = torch.bincount(perms[:, pos], minlength=33)
    expected = (1 << 17) / 33
    assert float((counts.double() - expected).abs().max()) < 6 * expected ** 0.5


def test_modinv_refuses_in_place(engine):
    """sc_modinv re-reads its operands on the error path: overlapping input and output are refused up front (round-2 advice)."""
    rng = random.Random(3)
    n = rng.getrandbits(1024) | (1 << 1023) | 1
    mod = engine.modulus(n)
    x = engine.upload([rng.randrange(2, n) for _ in range(64)], mod.nwords)
    with pytest.raises(ValueError, match="overlaps"):
        engine.modinv(mod, x, out=x)
    big = engine.upload([rng.randrange(2, n) for _ in range(8)] * 16, mod.nwords)
    with pytest.raises(ValueError, match="overlaps"):
        engine.modinv(mod, big[:100], out=big[28:128])


@pytest.mark.parametrize("pbits, dname, B, l", [
    (2048, "dgk_2048_l32", 65536, 32),       # BASELINE configs[2]: the headline
    (2048, "dgk_2048_l32", 131072, 32),      # per-GPU share of configs[3]: two to three rounds per pair launch, those stay whole
    (3072, "dgk_2048_l64", 32768, 64),       # per-GPU share of configs[4]: the L = 14 pair twins in eight / three segments (l14_rounds)
])
def test_the_configuration_bench_times(engine, keys, pbits, dname, B, l):
    """The shapes bench.py times, exactly as it runs them (the headline and the two `other_configs` / per-GPU shares): TWO concurrent
    shards (one library context, stream and host thread each, chip_share = 2 moving the one-lane threshold and cutting the long
    pair launches into segments), fixed-base tables built once and imported by the second context, step-4i shuffle on.
    Dec(result) == [x <= y] for all rows, and 20 sampled rows -- first and last of each shard, both sides of the shard cut, a
    partially filled wave -- bit-exact against the ORACLE (round 4 held the configs[3] / [4] shares with two shards to the decrypt
    property only)."""
    import bench
    from protocols.secure_comparison_amd import DGK, Paillier
    from protocols.secure_comparison_amd.batch import ConcurrentShards, PartySet, split_draws
    from protocols.secure_comparison_amd.distributed import shard_bounds
    from protocols.secure_comparison_amd.engine import Engine

    sk, dgk = oracle_paillier(keys, pbits), oracle_dgk(keys, dname)
    rbits, window = 400, min(bench.DEFAULT_FB_WINDOW, 20)
    engines = [Engine(), Engine()]
    sets = []
    for i, e in enumerate(engines):
        bob_p = Paillier(sk.n, sk.p, sk.q, engine=e)
        bob_d = DGK(dgk.n, dgk.g, dgk.h, dgk.u, dgk.t, dgk.p, dgk.q, dgk.v_p, dgk.v_q, engine=e, randomizer_bits=rbits, fixed_base_window=window)
        alice_d = bob_d.public_copy()
        if i > 0:
            bob_d.share_tables_from(sets[0].bob_dgk)
            alice_d.share_tables_from(sets[0].alice_dgk)
        alice_d.prepare(), bob_d.prepare()
        sets.append(PartySet(bob_p.public_copy(), alice_d, bob_p, bob_d, torch.cuda.Stream()))
    p0 = sets[0]
    x, y, x_enc, y_enc, draws = bench.synth_inputs(engines[0], l, p0.alice_paillier, p0.bob_paillier, p0.bob_dgk, B, rbits, seed=3, shuffle=True)
    bounds = [shard_bounds(B, i, 2) for i in range(2)]
    shard_inputs = [(x_enc[a:b].contiguous(), y_enc[a:b].contiguous(), d) for (a, b), d in zip(bounds, split_draws(draws, bounds))]
    torch.cuda.synchronize()
    runner = ConcurrentShards(sets)
    try:
        res = torch.cat(runner.run(shard_inputs, l, randomize=True), dim=0)
    finally:
        runner.close()
    dec = p0.bob_paillier.decrypt_raw_batch(res)
    assert bool(((dec[:, 0] == (x <= y).to(torch.int32)) & (dec[:, 1:] == 0).all(dim=1)).all().item())
    cut = bounds[0][1]
    idx = [0, 1, 7, 15, 16, 63, 64, 1000, cut - 65, cut - 2, cut - 1, cut, cut + 1, cut + 17, cut + 4096, B - 4097, B - 66, B - 3, B - 2, B - 1]
    assert engines[0].download(res[torch.tensor(idx, device=res.device)]) == _oracle_rows(engines[0], idx, l, sk, dgk, x_enc, y_enc, draws)
    del sets, shard_inputs, runner
    for e in engines:
        e.close()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("device_tensors", [True, False])
def test_interactive_batch_protocol_at_scale(engine, keys, device_tensors):
    """perform_secure_comparison_batch with draws=None at a BASELINE-shaped batch (l = 32, 2048/2048-bit keys, B = 8192): every
    random input drawn by the device generator, messages handed over on the device or through one pinned buffer; every result
    decrypts to [x <= y] and two runs give different ciphertexts."""
    sys.path.insert(0, os.path.dirname(__file__))
    import bench
    from _comm import DictionaryCommunicator
    from protocols.secure_comparison_amd import Initiator, KeyHolder, wire

    sk, dgk = oracle_paillier(keys, 2048), oracle_dgk(keys, "dgk_2048_l32")
    B, l = 8192, 32
    alice_p, alice_d, bob_p, bob_d = _schemes(engine, sk, dgk, 400)
    x, y, x_enc, y_enc, _ = bench.synth_inputs(engine, l, alice_p, bob_p, bob_d, B, 400, seed=9)
    outs = []
    for _ in range(2):
        box = {}
        alice = Initiator(l, DictionaryCommunicator(box, device_tensors), "bob")
        bob = KeyHolder(l, DictionaryCommunicator(box, device_tensors), "alice", bob_p, bob_d)
        wire.reset_stats()

        async def go():
            res, _ = await asyncio.gather(alice.perform_secure_comparison_batch(x_enc, y_enc, engine=engine), bob.perform_secure_comparison_batch())
            return res

        res = asyncio.run(go())
        dec = bob_p.decrypt_raw_batch(res)
        assert bool(((dec[:, 0] == (x <= y).to(torch.int32)) & (dec[:, 1:] == 0).all(dim=1)).all().item())
        assert (wire.STATS["bytes"] == 0) == device_tensors
        outs.append(res)
    assert not bool((outs[0] == outs[1]).all(dim=1).any().item())


def test_scheme_level_c_abi_with_ctypes_only(keys):
    """The scheme-level entry points of include/sc_amd.h driven the way INTEGRATION.md's stub drives them: ctypes, sc_malloc /
    sc_memcpy_*, no torch and none of this package's scheme classes.  A whole batch of comparisons is five library calls
    (sc_initiator_step1, sc_keyholder_step2_4b, sc_initiator_step4, sc_keyholder_step4j_5, sc_initiator_step67); every
    intermediate and the result are compared with the oracle bit for bit, for the key holder's CRT and literal paths."""
    import ctypes as C

    import numpy as np

    from oracle import sc_oracle as o
    from protocols.secure_comparison_amd import _lib
    from protocols.secure_comparison_amd.limbs import ints_to_words, words_to_ints

    lib = _lib.load()
    sk, dgk = oracle_paillier(keys, 1024), oracle_dgk(keys, "dgk_1024_l16")
    l, B, rbits = 16, 21, 400
    nw, nd, ew, er = 32, 32, (dgk.u.bit_length() + 31) // 32, (rbits + 31) // 32
    rng = random.Random(12)
    xs = [rng.randrange(1 << l) for _ in range(B)]
    ys = [xs[i] if i % 4 == 0 else rng.randrange(1 << l) for i in range(B)]
    drs = [o.draw(rng, l, sk, dgk, rbits) for _ in range(B)]
    x_enc = [sk.randomize(sk.enc_raw(x), 1 + rng.randrange(sk.n - 1)) for x in xs]
    y_enc = [sk.randomize(sk.enc_raw(y), 1 + rng.randrange(sk.n - 1)) for y in ys]
    traces = [dict() for _ in range(B)]
    expect = [o.compare(a, b, l, sk, dgk, d, True, t) for a, b, d, t in zip(x_enc, y_enc, drs, traces)]

    def words(v, w):
        a = ints_to_words([v], w)
        return a, a.ctypes.data_as(C.c_void_p)

    for flags in (0, 1):          # 1 = SC_KEY_NO_CRT
        ctx = C.c_void_p()
        assert lib.sc_ctx_create(0, C.byref(ctx)) == 0
        live = []

        def dev(host):
            p = C.c_void_p()
            assert lib.sc_malloc(ctx, max(host.nbytes, 4), C.byref(p)) == 0
            assert lib.sc_memcpy_h2d(ctx, p, host.ctypes.data_as(C.c_void_p), host.nbytes) == 0
            live.append(p)
            return p

        def empty(*shape, dtype=np.uint32):
            return dev(np.zeros(shape, dtype=dtype))

        def back(p, *shape, dtype=np.uint32):
            out = np.zeros(shape, dtype=dtype)
            assert lib.sc_memcpy_d2h(ctx, out.ctypes.data_as(C.c_void_p), p, out.nbytes) == 0
            return out

        def planes(rows, w):      # [l+1][B][w] bit-major from per-comparison lists
            return np.stack([ints_to_words([rows[b][i] for b in range(B)], w) for i in range(l + 1)])

        try:
            keep = [words(sk.n, nw), words(sk.p, nw // 2), words(sk.q, nw // 2)]
            a_key, b_key = C.c_int(), C.c_int()
            assert lib.sc_paillier_key_create(ctx, keep[0][1], nw, None, None, 0, 0, C.byref(a_key)) == 0, lib.sc_last_error(ctx)
            assert lib.sc_paillier_key_create(ctx, keep[0][1], nw, keep[1][1], keep[2][1], nw // 2, flags, C.byref(b_key)) == 0, lib.sc_last_error(ctx)
            dk = [words(dgk.n, nd), words(dgk.g, nd), words(dgk.h, nd), words(dgk.u, ew), words(dgk.p, nd // 2), words(dgk.q, nd // 2),
                  words(dgk.v_p, 5), words(dgk.v_q, 5)]
            a_dgk, b_dgk = C.c_int(), C.c_int()
            assert lib.sc_dgk_key_create(ctx, dk[0][1], dk[1][1], dk[2][1], nd, dk[3][1], ew, dgk.t, None, None, 0, None, None, 0, rbits, 8, 0,
                                         None, -1, C.byref(a_dgk)) == 0, lib.sc_last_error(ctx)
            assert lib.sc_dgk_key_create(ctx, dk[0][1], dk[1][1], dk[2][1], nd, dk[3][1], ew, dgk.t, dk[4][1], dk[5][1], nd // 2, dk[6][1], dk[7][1], 5,
                                         rbits, 8, flags, None, -1, C.byref(b_dgk)) == 0, lib.sc_last_error(ctx)
            # Alice: step 1 (+ randomization of [[z]])
            d_x, d_y = dev(ints_to_words(x_enc, 2 * nw)), dev(ints_to_words(y_enc, 2 * nw))
            d_r, d_rho_z = dev(ints_to_words([d.r for d in drs], nw)), dev(ints_to_words([d.rho_z for d in drs], nw))
            d_z, d_alpha, d_alpha_t, d_rsmall, d_rshift = empty(B, 2 * nw), empty(B, dtype=np.uint64), empty(B, dtype=np.uint64), \
                empty(B, dtype=np.uint64), empty(B, nw)
            assert lib.sc_initiator_step1(ctx, a_key, l, d_x, d_y, d_r, d_rho_z, 0, d_z, d_alpha, d_alpha_t, d_rsmall, d_rshift, B) == 0, lib.sc_last_error(ctx)
            assert words_to_ints(back(d_z, B, 2 * nw)) == [t["z_enc"] for t in traces]
            # Bob: steps 2, 4a, 4b (+ l + 1 randomizations)
            d_rb = dev(planes([[d.r_d] + d.r_beta for d in drs], er))
            d_zp, d_beta, d_dbit, d_z1, d_z2, d_db = empty(B, nw), empty(B, dtype=np.uint64), empty(B, dtype=np.uint64), empty(B, nw), \
                empty(B, nw), empty(l + 1, B, nd)
            assert lib.sc_keyholder_step2_4b(ctx, b_key, b_dgk, l, d_z, d_rb, er, 0, d_zp, d_beta, d_dbit, d_z1, d_z2, d_db, B) == 0, lib.sc_last_error(ctx)
            assert words_to_ints(back(d_zp, B, nw)) == [t["z"] for t in traces]
            got_db = back(d_db, l + 1, B, nd)
            assert words_to_ints(got_db[0]) == [t["d_sent"] for t in traces]
            assert [words_to_ints(got_db[1:, b]) for b in range(B)] == [t["beta_enc"] for t in traces]
            # Alice: steps 4c-4i (+ l + 1 randomizations, shuffle)
            rc_pre = [[None] * (l + 1) for _ in range(B)]      # the oracle randomizes output k = c_{perm[k]} after the shuffle
            for b, d in enumerate(drs):
                for k, src in enumerate(d.perm):
                    rc_pre[b][src] = d.r_c[k]
            d_rhos, d_ra = dev(planes([d.rhos for d in drs], ew)), dev(planes(rc_pre, er))
            d_perm = dev(np.array([d.perm for d in drs], dtype=np.int64))
            d_da = dev(np.array([d.delta_a for d in drs], dtype=np.uint64))
            d_c = empty(l + 1, B, nd)
            beta_ptr = C.c_void_p(d_db.value + B * nd * 4)      # [beta_i] = planes 1.. of the same array: no copy inside
            assert lib.sc_initiator_step4(ctx, a_dgk, l, d_db, beta_ptr, d_alpha, d_alpha_t, d_rsmall, d_da, d_rhos, ew, d_perm, d_ra, er, 0, None,
                                          d_c, B) == 0, lib.sc_last_error(ctx)
            got_c = back(d_c, l + 1, B, nd)
            assert [words_to_ints(got_c[:, b]) for b in range(B)] == [t["c_enc"] for t in traces]
            # Bob: steps 4j, 5 (+ 3 randomizations)
            d_rho3 = dev(np.concatenate([ints_to_words([getattr(d, f) for d in drs], nw) for f in ("rho_zeta1", "rho_zeta2", "rho_delta_b")]))
            d_delta_b, d_out3 = empty(B, dtype=np.uint64), empty(3, B, 2 * nw)
            assert lib.sc_keyholder_step4j_5(ctx, b_key, b_dgk, l, d_c, d_z1, d_z2, d_rho3, 0, d_delta_b, d_out3, B) == 0, lib.sc_last_error(ctx)
            assert back(d_delta_b, B, dtype=np.uint64).tolist() == [t["delta_b"] for t in traces]
            # Alice: steps 6, 7
            d_res = empty(B, 2 * nw)
            z1p, z2p, dbp = d_out3, C.c_void_p(d_out3.value + B * 2 * nw * 4), C.c_void_p(d_out3.value + 2 * B * 2 * nw * 4)
            assert lib.sc_initiator_step67(ctx, a_key, d_da, dbp, z1p, z2p, d_rsmall, d_rshift, 0, d_res, B) == 0, lib.sc_last_error(ctx)
            got = words_to_ints(back(d_res, B, 2 * nw))
            assert got == expect and [sk.dec_raw(v) for v in got] == [int(x <= y) for x, y in zip(xs, ys)]
            # the scheme objects' own operations: randomize, decrypt, encrypt bits, zero test -- one call each
            d_m = dev(ints_to_words(xs, nw))
            d_ct, d_rn, d_dec = empty(B, 2 * nw), empty(B, 2 * nw), empty(B, nw)
            assert lib.sc_paillier_encrypt(ctx, a_key, d_m, nw, 0, d_ct, B) == 0
            for key in (a_key, b_key):          # Alice's path (pairs modulo N) and the key holder's (CRT): identical integers
                assert lib.sc_paillier_randomize(ctx, key, d_ct, d_rho_z, d_rn, B) == 0, lib.sc_last_error(ctx)
                assert words_to_ints(back(d_rn, B, 2 * nw)) == [sk.randomize(sk.enc_raw(x), d.rho_z) for x, d in zip(xs, drs)]
            assert lib.sc_paillier_decrypt(ctx, b_key, d_rn, d_dec, B) == 0 and words_to_ints(back(d_dec, B, nw)) == xs
            assert lib.sc_paillier_decrypt(ctx, a_key, d_rn, d_dec, B) == -1 and b"secret" in lib.sc_last_error(ctx)
            bits = np.array([i & 1 for i in range(B)], dtype=np.uint8)
            d_bits, d_r1, d_e1, d_fl = dev(bits), dev(ints_to_words([d.r_d for d in drs], er)), empty(B, nd), empty(B, dtype=np.uint8)
            for key in (a_dgk, b_dgk):
                assert lib.sc_dgk_encrypt_bits_randomized(ctx, key, d_bits, d_r1, er, d_e1, B) == 0, lib.sc_last_error(ctx)
                assert words_to_ints(back(d_e1, B, nd)) == [dgk.randomize(dgk.enc_raw(int(b)), d.r_d) for b, d in zip(bits, drs)]
            assert lib.sc_dgk_is_zero(ctx, b_dgk, d_e1, d_fl, B) == 0 and back(d_fl, B, dtype=np.uint8).tolist() == [1 - (i & 1) for i in range(B)]
            assert lib.sc_dgk_is_zero(ctx, a_dgk, d_e1, d_fl, B) == -1
        finally:
            for p in live:
                lib.sc_free(ctx, p)
            lib.sc_ctx_destroy(ctx)


@pytest.mark.parametrize("B, shuffle", [(300, True), (9000, False)])
def test_randomizers_on_a_second_stream_give_the_same_residues(engine, keys, B, shuffle):
    """secure_comparison_batch(side=...): the 4 + 2(l+1) randomizer exponentiations run on a second library context and stream and
    are applied with one product each (SC_STEP_RANDOMIZERS_READY) -- every wire value and the result equal the fused path's."""
    import bench
    from protocols.secure_comparison_amd import DGK, Paillier
    from protocols.secure_comparison_amd.batch import BatchTrace, PartySet, secure_comparison_batch
    from protocols.secure_comparison_amd.engine import Engine

    sk, dgk = oracle_paillier(keys, 2048), oracle_dgk(keys, "dgk_2048_l32")
    l = 32
    alice_p, alice_d, bob_p, bob_d = _schemes(engine, sk, dgk, 400)
    e2 = Engine()
    bob_p2 = Paillier(sk.n, sk.p, sk.q, engine=e2)
    bob_d2 = DGK(dgk.n, dgk.g, dgk.h, dgk.u, dgk.t, dgk.p, dgk.q, dgk.v_p, dgk.v_q, engine=e2, randomizer_bits=400)
    alice_d2 = bob_d2.public_copy()
    bob_d2.share_tables_from(bob_d), alice_d2.share_tables_from(alice_d)
    side = PartySet(bob_p2.public_copy(), alice_d2, bob_p2, bob_d2, torch.cuda.Stream())
    x, y, x_enc, y_enc, draws = bench.synth_inputs(engine, l, alice_p, bob_p, bob_d, B, 400, seed=21, shuffle=shuffle)
    t1, t2 = BatchTrace(), BatchTrace()
    fused = secure_comparison_batch(x_enc, y_enc, l, alice_p, alice_d, bob_p, bob_d, draws, True, t1)
    split = secure_comparison_batch(x_enc, y_enc, l, alice_p, alice_d, bob_p, bob_d, draws, True, t2, side=side)
    torch.cuda.synchronize()
    for name in ("z_enc", "z", "d_enc", "beta_enc", "c_step4h", "c_sent", "delta_b", "zeta_1_enc", "zeta_2_enc", "delta_b_enc"):
        assert torch.equal(getattr(t1, name), getattr(t2, name)), name
    assert torch.equal(fused, split)
    dec = bob_p.decrypt_raw_batch(split)
    assert bool(((dec[:, 0] == (x <= y).to(torch.int32)) & (dec[:, 1:] == 0).all(dim=1)).all().item())
    idx = [0, 1, B // 2, B - 1]
    assert engine.download(split[torch.tensor(idx, device=engine.device)]) == _oracle_rows(engine, idx, l, sk, dgk, x_enc, y_enc, draws)
    e2.close()


def test_interactive_protocol_on_device_draws_is_bit_exact_for_sampled_rows(engine, keys):
    """The whole chain that nobody injects anything into: perform_secure_comparison_batch with draws=None at l = 32 / 2048-bit
    keys / B = 4096, the engine's generator keyed with a known key.  The draws of SAMPLED comparisons are restated on the CPU from
    the key alone (oracle/chacha_rng.py: every item has its own keystream; Alice's six generator calls come first, then the key
    holder's two) and fed to oracle.compare: the GPU's results for those rows must equal the oracle's bit for bit."""
    sys.path.insert(0, os.path.dirname(__file__))
    import bench
    from _comm import DictionaryCommunicator
    from oracle import sc_oracle as o
    from protocols.secure_comparison_amd import Initiator, KeyHolder

    sk, dgk = oracle_paillier(keys, 2048), oracle_dgk(keys, "dgk_2048_l32")
    B, l, rbits = 4096, 32, 400
    alice_p, alice_d, bob_p, bob_d = _schemes(engine, sk, dgk, rbits)
    x, y, x_enc, y_enc, _ = bench.synth_inputs(engine, l, alice_p, bob_p, bob_d, B, rbits, seed=33)
    box = {}
    alice = Initiator(l, DictionaryCommunicator(box), "bob")
    bob = KeyHolder(l, DictionaryCommunicator(box), "alice", bob_p, bob_d)

    async def go():
        res, _ = await asyncio.gather(alice.perform_secure_comparison_batch(x_enc, y_enc, engine=engine), bob.perform_secure_comparison_batch())
        return res

    engine.rng_seed(KEY)
    res = asyncio.run(go())
    dec = bob_p.decrypt_raw_batch(res)
    assert bool(((dec[:, 0] == (x <= y).to(torch.int32)) & (dec[:, 1:] == 0).all(dim=1)).all().item())
    idx = [0, 1, 63, 64, 511, 512, 513, B // 2, B - 65, B - 1]
    n, u, lp1 = sk.n, dgk.u, l + 1
    planes = lambda b: [i * B + b for i in range(lp1)]                                    # noqa: E731  (bit-major item indices)
    r = cr.rng_below(KEY, 0, n, B, items=idx)
    delta_a = cr.rng_coins(KEY, 1, B, items=idx)
    rhos = [cr.rng_below(KEY, 2, u, lp1 * B, nonzero=True, items=planes(b)) for b in idx]
    perms = cr.rng_permutations(KEY, 3, lp1, B, items=idx)
    rho_z = cr.rng_below(KEY, 4, n, B, nonzero=True, items=idx)
    r_alice = [cr.rng_bits(KEY, 5, rbits, lp1 * B, items=planes(b)) for b in idx]
    rho3 = [cr.rng_below(KEY, 6, n, 3 * B, nonzero=True, items=[b, B + b, 2 * B + b]) for b in idx]
    r_bob = [cr.rng_bits(KEY, 7, rbits, lp1 * B, items=planes(b)) for b in idx]
    xs, ys = engine.download(x_enc[torch.tensor(idx, device=engine.device)]), engine.download(y_enc[torch.tensor(idx, device=engine.device)])
    expect = []
    for k in range(len(idx)):
        pm = perms[k]
        rc = [r_alice[k][src] for src in pm]       # the library randomizes c_j with r_alice[j] before the shuffle, the oracle output k after it
        dr = o.Draws(r=r[k], delta_a=delta_a[k], rhos=rhos[k], perm=pm, rho_z=rho_z[k], r_d=r_bob[k][0], r_beta=r_bob[k][1:], r_c=rc,
                     rho_zeta1=rho3[k][0], rho_zeta2=rho3[k][1], rho_delta_b=rho3[k][2])
        expect.append(o.compare(xs[k], ys[k], l, sk, dgk, dr, True))
    assert engine.download(res[torch.tensor(idx, device=engine.device)]) == expect


def test_a_step_names_the_element_that_has_no_inverse(engine, keys):
    """A ciphertext that has no inverse modulo N^2 is reported by the step that inverts it as NotInvertibleError with its index in
    the batch (the verdict words are written by the inversion kernel into pinned host memory and read after one stream
    synchronisation: no copy kernel), and the engine works normally afterwards."""
    import bench
    from protocols.secure_comparison_amd.batch import secure_comparison_batch
    from protocols.secure_comparison_amd.engine import NotInvertibleError

    sk, dgk = oracle_paillier(keys, 1024), oracle_dgk(keys, "dgk_1024_l16")
    l, B = 16, 5000            # above the inversion tree's top: several levels
    alice_p, alice_d, bob_p, bob_d = _schemes(engine, sk, dgk, 400)
    x, y, x_enc, y_enc, draws = bench.synth_inputs(engine, l, alice_p, bob_p, bob_d, B, 400, seed=4, shuffle=True)
    good = secure_comparison_batch(x_enc, y_enc, l, alice_p, alice_d, bob_p, bob_d, draws)
    poisoned = x_enc.clone()
    poisoned[3777] = engine.upload([sk.p * 12345], x_enc.shape[-1])[0]          # gcd(x, N^2) = p
    for _ in range(2):
        with pytest.raises(NotInvertibleError) as ei:
            secure_comparison_batch(poisoned, y_enc, l, alice_p, alice_d, bob_p, bob_d, draws)
        assert ei.value.index == 3777
    again = secure_comparison_batch(x_enc, y_enc, l, alice_p, alice_d, bob_p, bob_d, draws)
    assert torch.equal(again, good)
    dec = bob_p.decrypt_raw_batch(again)
    assert bool((dec[:, 0] == (x <= y).to(torch.int32)).all().item())


@pytest.mark.parametrize("nbits", [2048, 2041])
def test_blinding_launch_in_the_modulus_multiple_context(engine, nbits):
    """sc_modexp_var above 8192 items of a (4,18) modulus runs modulo M = c n = -1 (mod 2^29) and leaves through the scaled
    reduction and the exact division by c (sc_device.h: redc_scaled / exact_div_small): results must be the canonical residues
    modulo n -- for edge operands (0, 1, n - 1, n - 2, values whose power is n - 1), random ones, with and without the fixed-base
    tail, in place and scattered.  A second modulus length moves the limb boundaries of M."""
    rng = random.Random(4100 + nbits)
    n = rng.getrandbits(nbits) | (1 << (nbits - 1)) | 1
    while (-pow(n, -1, 1 << 29)) % (1 << 29) == 1:        # (a modulus that is -1 already needs no multiple)
        n += 2
    mod = engine.modulus(n)
    count = 9001                                          # > 8192: not the small-batch configuration; a partially filled last wave
    xs = [0, 1, n - 1, n - 2, 2, (n - 1) // 2] + [rng.randrange(n) for _ in range(count - 6)]
    es = [1, 3, 1, 2, (1 << 34) - 1, 7] + [rng.randrange(1, 1 << 34) for _ in range(count - 6)]
    es[2] = 1                                             # (n - 1)^1 = n - 1: the largest canonical result, X = c (n - 1)
    es[3] = 2
    tx, te = engine.upload(xs, mod.nwords), engine.upload(es, 2)
    assert engine.download(engine.modexp_var(mod, tx, te, 34)) == [pow(x, e, n) for x, e in zip(xs, es)]
    h = rng.randrange(2, n)
    fb = engine.fixed_base(mod, h, 400, window=8)
    rs = [0, 1, (1 << 400) - 1] + [rng.getrandbits(400) for _ in range(count - 3)]
    tr = engine.upload(rs, 13)
    expect = [pow(x, e, n) * pow(h, r, n) % n for x, e, r in zip(xs, es, rs)]
    assert engine.download(engine.modexp_var(mod, tx, te, 34, fb, tr)) == expect
    perm = list(range(count))
    rng.shuffle(perm)
    dest = torch.tensor(perm, dtype=torch.int64, device=engine.device)
    got = engine.download(engine.modexp_var(mod, tx, te, 34, fb, tr, dest=dest))
    assert [got[perm[i]] for i in range(count)] == expect
    # the same numbers below the threshold (small-batch configuration, original modulus): equal results
    small = 500
    assert engine.download(engine.modexp_var(mod, tx[:small].contiguous(), te[:small].contiguous(), 34, fb, tr[:small].contiguous())) == expect[:small]
