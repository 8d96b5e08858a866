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


def test_the_configuration_bench_times(engine, keys):
    """BASELINE configs[2] exactly as bench.py runs it: B = 65536, l = 32, 2048/2048-bit keys, TWO concurrent shards (one
    library context, stream and host thread each, chip_share = 2 moving the one-lane threshold), fixed-base tables built once
    and imported by the second context, step-4i shuffle on.  Dec(result) == [x <= y] for all rows, and 20 sampled rows -- first
    and last of each shard, both sides of the shard cut, a partially filled wave -- bit-exact against the oracle."""
    import bench
    from protocols.secure_comparison_amd import DGK, Paillier
    from protocols.secure_comparison_amd.batch import ConcurrentShards, PartySet, split_draws
    from protocols.secure_comparison_amd.distributed import shard_bounds
    from protocols.secure_comparison_amd.engine import Engine

    sk, dgk = oracle_paillier(keys, 2048), oracle_dgk(keys, "dgk_2048_l32")
    B, l, rbits, window = 65536, 32, 400, bench.DEFAULT_FB_WINDOW
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
    for e in engines:
        e.close()


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
