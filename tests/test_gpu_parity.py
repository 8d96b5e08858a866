"""GPU parity tests proper: every call goes through the C ABI (libsc_amd.so) on a real MI355X and is compared bit
for bit with the CPU oracle / the committed golden fixtures.  Bar: bit-exact (integer work)."""
import asyncio
import json
import os
import random
import warnings

import pytest
import torch

from conftest import GOLDEN, oracle_dgk, oracle_paillier
from oracle import sc_oracle as o

pytestmark = pytest.mark.gpu
H = lambda s: int(s, 16)  # noqa: E731


# ------------------------------------------------------------------------------------------ primitives
def test_primitive_kats_all_sizes(engine):
    """Known-answer tests at 128 / 1024 / 2048 / 3072 / 4096 / 6144 bits incl. operands 0, 1, n-1 (every kernel configuration)."""
    for k in json.load(open(os.path.join(GOLDEN, "kat_primitives.json"))):
        n = H(k["n"])
        mod = engine.modulus(n)
        a, b = engine.upload([H(x) for x in k["a"]], mod.nwords), engine.upload([H(x) for x in k["b"]], mod.nwords)
        assert engine.download(engine.modmul(mod, a, b)) == [H(x) for x in k["mul"]], k["bits"]
        assert engine.download(engine.modexp_shared(mod, a, H(k["e"]))) == [H(x) for x in k["pow"]], k["bits"]
        inv = engine.modinv(mod, engine.upload([H(x) for x in k["inv_in"]], mod.nwords))
        assert engine.download(inv) == [H(x) for x in k["inv"]], k["bits"]
        es = [H(x) for x in k["small_e"]]
        base = engine.upload([H(k["a"][5])] * len(es), mod.nwords)
        got = engine.download(engine.modexp_var(mod, base, engine.upload(es, 3), 67))
        assert got == [H(x) for x in k["pow_small"]], k["bits"]


@pytest.mark.parametrize("count", [0, 1, 15, 17, 63, 65, 333])
def test_ragged_and_empty_batches(engine, count):
    rng = random.Random(count)
    n = rng.getrandbits(2048) | (1 << 2047) | 1
    mod = engine.modulus(n)
    a = [rng.randrange(n) for _ in range(count)]
    b = [rng.randrange(n) for _ in range(count)]
    ta, tb = engine.upload(a, mod.nwords).reshape(count, mod.nwords), engine.upload(b, mod.nwords).reshape(count, mod.nwords)
    assert engine.download(engine.modmul(mod, ta, tb)) == [x * y % n for x, y in zip(a, b)]
    assert engine.download(engine.modexp_shared(mod, ta, 65537)) == [pow(x, 65537, n) for x in a]
    if count:
        inv_in = [x for x in a if _invertible(x, n)]
        got = engine.download(engine.modinv(mod, engine.upload(inv_in, mod.nwords)))
        assert got == [pow(x, -1, n) for x in inv_in]


def _invertible(x, n):
    try:
        pow(x, -1, n)
        return True
    except ValueError:
        return False


def test_fixed_base_windows_and_fused_forms(engine):
    rng = random.Random(3)
    n = rng.getrandbits(2048) | (1 << 2047) | 1
    mod = engine.modulus(n)
    h = rng.randrange(2, n)
    B = 70
    c = [rng.randrange(n) for _ in range(B)]
    r = [rng.getrandbits(400) for _ in range(B)]
    r[0], r[1], r[2] = 0, 1, (1 << 400) - 1
    rho = [1 + rng.randrange((1 << 34) - 1) for _ in range(B)]
    tc, tr, trho = engine.upload(c, mod.nwords), engine.upload(r, 13), engine.upload(rho, 2)
    for window in (1, 5, 8, 11):
        fb = engine.fixed_base(mod, h, 400, window)
        assert engine.download(engine.fixedbase_pow(fb, tr)) == [pow(h, x, n) for x in r]
        assert engine.download(engine.fixedbase_pow(fb, tr, mul_into=tc)) == [pow(h, x, n) * y % n for x, y in zip(r, c)]
        got = engine.download(engine.modexp_var(mod, tc, trho, 34, fb, tr))
        assert got == [pow(y, e, n) * pow(h, x, n) % n for y, e, x in zip(c, rho, r)]


def test_wide_operand_reduction_and_is_one(engine, keys):
    d = oracle_dgk(keys, "dgk_2048_l32")
    mp = engine.modulus(d.p)
    rng = random.Random(9)
    ms = [0 if i % 3 == 0 else rng.randrange(1, d.u) for i in range(50)]
    cts = [d.randomize(d.enc_raw(m), rng.getrandbits(400)) for m in ms]
    t = engine.upload(cts, (d.n.bit_length() + 31) // 32)
    assert engine.modexp_shared_isone(mp, t, d.v_p).tolist() == [int(m == 0) for m in ms]
    assert engine.download(engine.modexp_shared(mp, t, 1)) == [c % d.p for c in cts]


def test_errors(engine):
    from protocols.secure_comparison_amd.engine import NotInvertibleError

    n = 3 * 5 * 7 * (random.Random(1).getrandbits(1000) | 1)
    mod = engine.modulus(n)
    with pytest.raises(NotInvertibleError):
        engine.modinv(mod, engine.upload([2, 21, 4], mod.nwords))
    many = [2] * 500 + [35] + [4] * 500
    with pytest.raises(NotInvertibleError):
        engine.modinv(mod, engine.upload(many, mod.nwords))
    with pytest.raises(ValueError):
        engine.modulus(1 << 64)            # even modulus
    with pytest.raises(Exception):
        engine.modulus((1 << 9000) + 1)    # larger than any compiled configuration


# ------------------------------------------------------------------------------------------ scheme level
@pytest.mark.parametrize("bits", [1024, 2048, 3072])
def test_paillier_pieces(engine, keys, bits):
    from protocols.secure_comparison_amd import Paillier

    sk = oracle_paillier(keys, bits)
    rng = random.Random(bits)
    B = 40
    ms = [0, 1, sk.n - 1] + [rng.randrange(sk.n) for _ in range(B - 3)]
    rhos = [1 + rng.randrange(sk.n - 1) for _ in range(B)]
    for use_crt in (False, True):
        p = Paillier(sk.n, sk.p, sk.q, engine=engine, use_crt=use_crt)
        nw = p.mod_n.nwords
        enc = p.encrypt_raw_batch(engine.upload(ms, nw))
        assert engine.download(enc) == [sk.enc_raw(m) for m in ms]
        rnd = p.randomize_batch(enc, engine.upload(rhos, nw))
        assert engine.download(rnd) == [sk.randomize(sk.enc_raw(m), r) for m, r in zip(ms, rhos)]
        assert engine.download(p.decrypt_raw_batch(rnd)) == ms
    wide = [m + (1 << 64) for m in ms]     # 2^l + r may exceed N (SC/initiator.py:256)
    assert engine.download(p.encrypt_raw_batch(engine.upload(wide, nw + 1))) == [sk.enc_raw(m) for m in wide]


@pytest.mark.parametrize("l", [16, 32, 64])
def test_plaintext_side_kernels(engine, keys, l):
    sk = oracle_paillier(keys, 2048)
    n = sk.n
    rng = random.Random(l)
    rs = [0, 1, n - 1, (n - 1) // 2, (n - 1) // 2 - 1, (n - 1) // 2 + 1, (1 << l) - 1, 1 << l] + [rng.randrange(n) for _ in range(92)]
    t = engine.upload(rs, 64)
    M = (1 << 64) - 1
    m1, al, at, rsm, rsh = engine.plain_alice(t, n, l)
    assert engine.download(m1) == [(1 << l) + r for r in rs]
    assert [v & M for v in al.tolist()] == [r % (1 << l) for r in rs]
    assert [v & M for v in at.tolist()] == [(r - n) % (1 << l) for r in rs]
    assert rsm.tolist() == [int(r < (n - 1) // 2) for r in rs]
    assert engine.download(rsh) == [r >> l for r in rs]
    be, db, z1, z2 = engine.plain_bob(t, n, l)
    assert [v & M for v in be.tolist()] == [r % (1 << l) for r in rs]
    assert db.tolist() == [int(r < (n - 1) // 2) for r in rs]
    assert engine.download(z1) == [r >> l for r in rs]
    assert engine.download(z2) == [((r + n) >> l) if r < (n - 1) // 2 else (r >> l) for r in rs]


# ------------------------------------------------------------------------------------------ whole comparisons
def _schemes(engine, sk, dgk, rbits, use_crt=True):
    from protocols.secure_comparison_amd import DGK, Paillier

    bob_p = Paillier(sk.n, sk.p, sk.q, engine=engine, use_crt=use_crt)
    bob_d = DGK(dgk.n, dgk.g, dgk.h, dgk.u, dgk.t, dgk.p, dgk.q, dgk.v_p, dgk.v_q, engine=engine, randomizer_bits=rbits)
    return bob_p.public_copy(), bob_d.public_copy(), bob_p, bob_d


def _draw_tensors(engine, drs, l, nw, ew, er, device):
    from protocols.secure_comparison_amd.batch import BatchDraws

    B = len(drs)
    bm = lambda rows, w: torch.stack([engine.upload([rows[b][i] for b in range(B)], w) for i in range(l + 1)])  # noqa: E731
    has_perm = drs[0].perm is not None
    rc = [list(d.r_c) for d in drs]
    if has_perm:
        rc = [[None] * (l + 1) for _ in range(B)]
        for b, d in enumerate(drs):
            for k, src in enumerate(d.perm):
                rc[b][src] = d.r_c[k]
    return BatchDraws(r=engine.upload([d.r for d in drs], nw), delta_a=engine.upload_u64([d.delta_a for d in drs]),
                      rhos=bm([d.rhos for d in drs], ew),
                      permutation=torch.tensor([d.perm for d in drs], dtype=torch.int64, device=device) if has_perm else None,
                      rho_z=engine.upload([d.rho_z for d in drs], nw), r_bob_dgk=bm([[d.r_d] + d.r_beta for d in drs], er),
                      r_alice_dgk=bm(rc, er), rho_zeta_1=engine.upload([d.rho_zeta1 for d in drs], nw),
                      rho_zeta_2=engine.upload([d.rho_zeta2 for d in drs], nw), rho_delta_b=engine.upload([d.rho_delta_b for d in drs], nw))


@pytest.mark.parametrize("set_index", [0, 1, 2])
def test_golden_comparisons_through_hip(engine, keys, set_index):
    """The committed golden comparisons (inputs, every random draw, wire values, result) reproduced by the HIP path."""
    from protocols.secure_comparison_amd.batch import BatchTrace, secure_comparison_batch
    from test_oracle_golden import load_draws

    s = json.load(open(os.path.join(GOLDEN, "comparisons.json")))[set_index]
    sk, dgk = oracle_paillier(keys, s["paillier_bits"]), oracle_dgk(keys, s["dgk"])
    alice_p, alice_d, bob_p, bob_d = _schemes(engine, sk, dgk, s["rbits"])
    items, l = s["items"], s["l"]
    drs = [load_draws(it["draws"]) for it in items]
    nw = bob_p.mod_n.nwords
    draws = _draw_tensors(engine, drs, l, nw, (dgk.u.bit_length() + 31) // 32, (s["rbits"] + 31) // 32, engine.device)
    tr = BatchTrace()
    res = secure_comparison_batch(engine.upload([H(it["x_enc"]) for it in items], 2 * nw), engine.upload([H(it["y_enc"]) for it in items], 2 * nw),
                                  l, alice_p, alice_d, bob_p, bob_d, draws, True, tr)
    assert engine.download(res) == [H(it["result"]) for it in items]
    assert engine.download(tr.z_enc) == [H(it["z_enc"]) for it in items]
    assert engine.download(tr.z) == [H(it["z"]) for it in items]
    assert tr.delta_b.tolist() == [it["delta_b"] for it in items]
    for b, it in enumerate(items):
        assert engine.download(tr.c_sent[:, b]) == [H(x) for x in it["c_sent"]]
    static = secure_comparison_batch(engine.upload([H(it["x_enc"]) for it in items], 2 * nw), engine.upload([H(it["y_enc"]) for it in items], 2 * nw),
                                     l, alice_p, alice_d, bob_p, bob_d, draws, False)
    assert engine.download(static) == [H(it["result_static"]) for it in items]


@pytest.mark.parametrize("pbits, dname, B, rbits, use_crt", [
    (1024, "dgk_1024_l16", 40, 400, True), (1024, "dgk_1024_l16", 20, 400, False),
    (2048, "dgk_2048_l32", 20, 400, True), (2048, "dgk_2048_l64", 6, 400, True), (3072, "dgk_3072_l64", 3, 400, True)])
def test_random_batches_vs_oracle(engine, keys, pbits, dname, B, rbits, use_crt):
    """Seeded random comparisons incl. equal / adjacent pairs; l = 64 exercises 67-bit blinding exponents and 3072-bit keys
    (N^2 = 6144 bits) the largest kernel configurations (BASELINE configs[4] shape)."""
    from protocols.secure_comparison_amd.batch import secure_comparison_batch

    sk, dgk = oracle_paillier(keys, pbits), oracle_dgk(keys, dname)
    l = keys[dname]["l"]
    alice_p, alice_d, bob_p, bob_d = _schemes(engine, sk, dgk, rbits, use_crt)
    rng = random.Random(pbits * 1000 + l)
    xs = [rng.randrange(1 << l) for _ in range(B)]
    ys = [xs[i] if i % 4 == 0 else (max(xs[i] - 1, 0) if i % 4 == 1 else rng.randrange(1 << l)) for i in range(B)]
    drs = [o.draw(rng, l, sk, dgk, rbits) for _ in range(B)]
    x_enc = [sk.randomize(sk.enc_raw(x), 1 + rng.randrange(sk.n - 1)) for x in xs]
    y_enc = [sk.randomize(sk.enc_raw(y), 1 + rng.randrange(sk.n - 1)) for y in ys]
    expect = [o.compare(a, b, l, sk, dgk, d, True) for a, b, d in zip(x_enc, y_enc, drs)]
    nw = bob_p.mod_n.nwords
    draws = _draw_tensors(engine, drs, l, nw, (dgk.u.bit_length() + 31) // 32, (rbits + 31) // 32, engine.device)
    got = engine.download(secure_comparison_batch(engine.upload(x_enc, 2 * nw), engine.upload(y_enc, 2 * nw), l, alice_p, alice_d,
                                                  bob_p, bob_d, draws))
    assert got == expect
    assert [sk.dec_raw(c) for c in got] == [int(x <= y) for x, y in zip(xs, ys)]


def test_config2_full_size_property(engine, keys):
    """BASELINE configs[1]: batch 4096, l = 16, 2048-bit keys.  Size-independent properties: Dec(result) == [x <= y] for every
    comparison (decrypted on the GPU), and a sampled subset bit-exact against the oracle."""
    import bench

    sk, dgk = oracle_paillier(keys, 2048), oracle_dgk(keys, "dgk_2048_l16")
    alice_p, alice_d, bob_p, bob_d = _schemes(engine, sk, dgk, 400)
    from protocols.secure_comparison_amd.batch import secure_comparison_batch

    B, l = 4096, 16
    x, y, x_enc, y_enc, draws = bench.synth_inputs(engine, l, alice_p, bob_p, bob_d, B, 400, seed=0)
    res = secure_comparison_batch(x_enc, y_enc, l, alice_p, alice_d, bob_p, bob_d, draws)
    dec = bob_p.decrypt_raw_batch(res)
    assert bool(((dec[:, 0] == (x <= y).to(torch.int32)) & (dec[:, 1:] == 0).all(dim=1)).all().item())
    idx = [0, 1, 2, 7, 8, 9, 1234, 4095]
    ints = lambda t: engine.download(t[idx])  # noqa: E731
    perb = lambda t: [engine.download(t[:, i]) for i in idx]  # noqa: E731
    M = (1 << 64) - 1
    rows = zip(ints(x_enc), ints(y_enc), ints(draws.r), [int(v) & M for v in draws.delta_a[idx].tolist()], perb(draws.rhos),
               ints(draws.rho_z), perb(draws.r_bob_dgk), perb(draws.r_alice_dgk), ints(draws.rho_zeta_1), ints(draws.rho_zeta_2),
               ints(draws.rho_delta_b))
    expect = []
    for xe, ye, r, da, rhos, rho_z, rb, rc, z1, z2, zb in rows:
        dr = o.Draws(r=r, delta_a=da, rhos=rhos, perm=None, rho_z=rho_z, r_d=rb[0], r_beta=rb[1:], r_c=rc, rho_zeta1=z1,
                     rho_zeta2=z2, rho_delta_b=zb)
        expect.append(o.compare(xe, ye, l, sk, dgk, dr, True))
    assert ints(res) == expect


# ------------------------------------------------------------------------------------------ reference-style object API on the GPU
def test_object_api_reference_identities(engine, keys):
    """A condensed replay of the reference's unit tests (test_secure_comparison.py:196-635) on the product's own classes."""
    from protocols.secure_comparison_amd import DGK, Initiator, KeyHolder, Paillier, PaillierCiphertext, to_bits

    sk, od = oracle_paillier(keys, 1024), oracle_dgk(keys, "dgk_tiny_l16")
    paillier = Paillier(sk.n, sk.p, sk.q, engine=engine)
    dgk_full = DGK(od.n, od.g, od.h, od.u, od.t, od.p, od.q, od.v_p, od.v_q, full_decryption=True, engine=engine, randomizer_bits=50)
    l, n, u = 16, sk.n, od.u
    # test_modulo_n_squared (:166-182)
    y_enc = PaillierCiphertext(paillier.public_key.n_squared - 2, paillier)
    y = paillier.decrypt(y_enc, apply_encoding=False)
    z_enc = y_enc - paillier.unsafe_encrypt(0, apply_encoding=False) + paillier.unsafe_encrypt((1 << l) + n * n, apply_encoding=False)
    assert (y + (1 << l) + n * n) % (n * n) == paillier.decrypt(z_enc, apply_encoding=False)
    for x, y in ((-400, -383), (230, 269), (8668, 9015)):
        x_enc, y_enc = paillier.unsafe_encrypt(x), paillier.unsafe_encrypt(y)
        z_enc, r = Initiator.step_1(x_enc, y_enc, l, paillier)
        assert paillier.decrypt(paillier.unsafe_encrypt((y - x + (1 << l) + r) % (n * n), apply_encoding=False)) == paillier.decrypt(z_enc)
        z, beta = KeyHolder.step_2(z_enc, l, paillier)
        alpha = Initiator.step_3(r, l)
        d_enc = Initiator.step_4c(KeyHolder.step_4a(z, dgk_full, paillier, l), r, dgk_full, paillier)
        beta_is_enc = KeyHolder.step_4b(beta, l, dgk_full)
        assert [dgk_full.decrypt(b) for b in beta_is_enc] == to_bits(beta, l)
        xor = Initiator.step_4d(alpha, beta_is_enc)
        w_is_enc, alpha_tilde = Initiator.step_4e(r, alpha, xor, d_enc, paillier)
        d = dgk_full.decrypt(d_enc)
        beta_bits = to_bits(beta, l)
        w_plain = [((alpha[i] ^ beta_bits[i]) - (0 if alpha[i] == alpha_tilde[i] else d)) % u for i in range(l)]
        assert [dgk_full.decrypt(w, apply_encoding=False) for w in w_is_enc] == w_plain
        w_is_enc = Initiator.step_4f(w_is_enc)
        w_plain = [w * 2 ** i % u for i, w in enumerate(w_plain)]
        assert [dgk_full.decrypt(w, apply_encoding=False) for w in w_is_enc] == w_plain
        s, delta_a = Initiator.step_4g()
        c_is_enc = Initiator.step_4h(s, alpha, alpha_tilde, d_enc, beta_is_enc, w_is_enc, delta_a, dgk_full)
        for i in range(l):
            t = s + alpha[i] + (alpha_tilde[i] - alpha[i]) * d - beta_bits[i] + 3 * sum(w_plain[i + 1:])
            assert t % u == dgk_full.decrypt(c_is_enc[i + 1], apply_encoding=False)
        c_is_enc = Initiator.step_4i(c_is_enc, dgk_full)
        delta_b = KeyHolder.step_4j(c_is_enc, dgk_full)
        assert delta_b == int(any(dgk_full.is_zero(c) for c in c_is_enc))
        zeta_1, zeta_2, delta_b_enc = KeyHolder.step_5(z, l, delta_b, paillier)
        res = Initiator.step_7(zeta_1, zeta_2, r, l, Initiator.step_6(delta_a, delta_b_enc), paillier)
        assert paillier.decrypt(res) == int(x <= y)
    for z, delta_b in ((n, 0), (100, 1)):      # test_step_5 (:548-566)
        z1, z2, db = KeyHolder.step_5(z, l, delta_b, paillier)
        assert paillier.decrypt(z1, apply_encoding=False) == (z >> l) % n and paillier.decrypt(db) == delta_b
        assert paillier.decrypt(z2, apply_encoding=False) == (((z + n) >> l) if z < (n - 1) // 2 else (z >> l)) % n


def test_interactive_protocol_on_gpu_strict(engine, keys):
    """Both players over an in-memory transport with warnings-as-errors (the reference's strict fixtures,
    test/conftest.py:26-36; flows of test_secure_comparison.py:641-835)."""
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    from _comm import DictionaryCommunicator
    from protocols.secure_comparison_amd import DGK, Initiator, KeyHolder, Paillier

    sk, od = oracle_paillier(keys, 1024), oracle_dgk(keys, "dgk_1024_l16")
    bob_p = Paillier(sk.n, sk.p, sk.q, engine=engine)
    bob_d = DGK(od.n, od.g, od.h, od.u, od.t, od.p, od.q, od.v_p, od.v_q, engine=engine)
    box = {}
    alice = Initiator(16, DictionaryCommunicator(box), "bob")
    bob = KeyHolder(16, DictionaryCommunicator(box), "alice", bob_p, bob_d)

    async def go(x, y):
        res, _ = await asyncio.gather(alice.perform_secure_comparison(x, y), bob.perform_secure_comparison())
        return res

    with warnings.catch_warnings():
        warnings.filterwarnings("error", ".*ciphertext", UserWarning)
        warnings.filterwarnings("error", ".*randomness", UserWarning)
        for x, y in ((23, 42), (42, 23), (9, 9)):
            assert bob_p.decrypt(asyncio.run(go(x, y))) == int(x <= y)
    assert alice.scheme_paillier == bob_p and alice.scheme_dgk == bob_d
    bob_p.shut_down(), bob_d.shut_down()


def test_key_generation_and_fresh_keys(engine):
    """from_security_parameter (SC/keyholder.py:156-166) with small sizes: fresh keys work end to end on the GPU."""
    from protocols.secure_comparison_amd import DGK, Initiator, KeyHolder, Paillier
    from protocols.secure_comparison_amd.keygen import next_prime

    pai = Paillier.from_security_parameter(key_length=512, engine=engine)
    dgk = DGK.from_security_parameter(v_bits=40, n_bits=512, u=next_prime(1 << 10), full_decryption=True, engine=engine)
    assert dgk.decrypt(dgk.encrypt(5) + dgk.encrypt(-2)) == 3 and pai.decrypt(pai.encrypt(-7) * 3) == -21
    z_enc, r = Initiator.step_1(pai.unsafe_encrypt(3), pai.unsafe_encrypt(4), 8, pai)
    assert KeyHolder.step_2(z_enc, 8, pai)[0] == (4 - 3 + 256 + r) % pai.public_key.n


def test_batched_interactive_protocol_on_gpu(engine, keys):
    """Both players' perform_secure_comparison_batch over the in-memory transport, arrays handed over on the device and
    serialized through one pinned host buffer: injected draws -> bit-exact vs the oracle; draws=None -> every random input
    from the device generator, correct bits."""
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    from _comm import DictionaryCommunicator
    from protocols.secure_comparison_amd import Initiator, KeyHolder

    sk, dgk = oracle_paillier(keys, 1024), oracle_dgk(keys, "dgk_1024_l16")
    l, B, rbits = 16, 37, 400
    _, _, bob_p, bob_d = _schemes(engine, sk, dgk, rbits)
    rng = random.Random(8)
    xs = [rng.randrange(1 << l) for _ in range(B)]
    ys = [xs[i] if i % 3 == 0 else rng.randrange(1 << l) for i in range(B)]
    x_enc = [sk.randomize(sk.enc_raw(x), 1 + rng.randrange(sk.n - 1)) for x in xs]
    y_enc = [sk.randomize(sk.enc_raw(y), 1 + rng.randrange(sk.n - 1)) for y in ys]
    drs = [o.draw(rng, l, sk, dgk, rbits) for _ in range(B)]
    nw = bob_p.mod_n.nwords
    draws = _draw_tensors(engine, drs, l, nw, (dgk.u.bit_length() + 31) // 32, (rbits + 31) // 32, engine.device)
    tx, ty = engine.upload(x_enc, 2 * nw), engine.upload(y_enc, 2 * nw)
    for use_draws, device_tensors in ((True, True), (True, False), (False, True), (False, False)):
        box = {}
        alice = Initiator(l, DictionaryCommunicator(box, device_tensors), "bob")
        bob = KeyHolder(l, DictionaryCommunicator(box, device_tensors), "alice", bob_p, bob_d)

        async def go():
            res, _ = await asyncio.gather(alice.perform_secure_comparison_batch(tx, ty, draws if use_draws else None, engine=engine),
                                          bob.perform_secure_comparison_batch(draws if use_draws else None))
            return res

        with warnings.catch_warnings():
            warnings.filterwarnings("error", ".*randomness", UserWarning)
            got = engine.download(asyncio.run(go()))
        assert [sk.dec_raw(v) for v in got] == [int(x <= y) for x, y in zip(xs, ys)]
        if use_draws:
            assert got == [o.compare(a, b, l, sk, dgk, d, True) for a, b, d in zip(x_enc, y_enc, drs)]


@pytest.mark.parametrize("bits", [512, 1024, 1536, 1600, 2048, 3072, 3200, 4096, 6144, 6400, 6470, 8192, 8330])
def test_worst_case_limbs(engine, bits):
    """Moduli and operands whose 29-bit limbs are all ones (n = 2^bits - c, a = n - 1, n - 2, 2^k - 1): the largest column
    sums the lazy 64-bit accumulators can see, in every kernel configuration (1600 / 3200 / 6400 bits select the L = 14
    family, 1536 / 3072 / 6144 the L = 27 one), through products, squarings and the wide-operand reduction.  6470 and 8330 bits
    are the two shapes whose residue arrays (203 / 261 words) fill all 29 * S bits of their configuration, so a raw chunk of the
    wide operand can reach R - 1 and the lazy sum in the Horner reduction exceeds R (found by tools/gpu_fuzz.py)."""
    n = (1 << bits) - 1
    while n % 3 == 0 or n % 5 == 0 or n % 2 == 0:
        n -= 2
    mod = engine.modulus(n)
    vals = [n - 1, n - 2, (1 << (bits - 1)) - 1, (n - 1) // 2, n - (1 << 29), 1, 0, (1 << (bits - 3)) + 1]
    t = engine.upload(vals, mod.nwords)
    e = (1 << 200) - 1          # all-ones exponent: squarings and products alternate densely
    assert engine.download(engine.modexp_shared(mod, t, e)) == [pow(v, e, n) for v in vals]
    assert engine.download(engine.modmul(mod, t, t)) == [v * v % n for v in vals]
    assert engine.download(engine.modexp_shared(mod, t, 2, mul_into=t)) == [pow(v, 3, n) for v in vals]
    wide = [(1 << (2 * bits)) - 1 - 977 * i for i in range(8)]     # all-ones operand twice as wide as the modulus
    assert engine.download(engine.modexp_shared(mod, engine.upload(wide, 2 * mod.nwords), 3)) == [pow(w, 3, n) for w in wide]
    ev = [(1 << 67) - 1, (1 << 66) + 12345, 7, 1, 0, (1 << 35) - 1, 3, 2]
    assert engine.download(engine.modexp_var(mod, t, engine.upload(ev, 3), 67)) == [pow(v, x, n) for v, x in zip(vals, ev)]


def test_steps_6_7_fused_equals_separate(engine, keys):
    """Initiator.step_6_7_batch (one inversion) gives the same residues as step_6_batch followed by step_7_batch (two), and
    sc_paillier_encrypt_raw_neg equals the modular inverse of sc_paillier_encrypt_raw (incl. m = 0)."""
    from protocols.secure_comparison_amd import Initiator, Paillier

    sk = oracle_paillier(keys, 2048)
    pai = Paillier(sk.n, engine=engine)
    rng = random.Random(12)
    B, l, nw = 50, 32, pai.mod_n.nwords
    ms = [0, 1, sk.n - 1] + [rng.randrange(sk.n) for _ in range(B - 3)]
    neg = pai.encrypt_raw_neg_batch(engine.upload(ms, nw))
    assert engine.download(neg) == [sk.neg(sk.enc_raw(m)) for m in ms]
    rs = [rng.randrange(sk.n) for _ in range(B)]
    rs[0], rs[1] = 5, sk.n - 5
    _, plain = Initiator.step_1_batch(engine.upload([sk.enc_raw(1)] * B, 2 * nw), engine.upload([sk.enc_raw(2)] * B, 2 * nw), l, pai,
                                      engine.upload(rs, nw))
    rnd = lambda m: sk.randomize(sk.enc_raw(m), 1 + rng.randrange(sk.n - 1))  # noqa: E731
    da = [rng.randrange(2) for _ in range(B)]
    db, z1, z2 = [rnd(rng.randrange(2)) for _ in range(B)], [rnd(rng.randrange(1 << 40)) for _ in range(B)], [rnd(rng.randrange(1 << 40)) for _ in range(B)]
    tda, tdb, tz1, tz2 = engine.upload_u64(da), engine.upload(db, 2 * nw), engine.upload(z1, 2 * nw), engine.upload(z2, 2 * nw)
    sep = Initiator.step_7_batch(tz1, tz2, plain, l, Initiator.step_6_batch(tda, tdb, pai), pai)
    fused = Initiator.step_6_7_batch(tda, tdb, tz1, tz2, plain, l, pai)
    expect = [o.step_7(a, b, r, l, o.step_6(d, c, sk), sk) for a, b, r, d, c in zip(z1, z2, rs, da, db)]
    assert engine.download(sep) == expect and engine.download(fused) == expect


def test_table_traffic_probe_is_identity(engine):
    """sc_table_traffic_probe (the FETCH_SIZE / WRITE_SIZE calibration launch) writes and re-reads table rows and must hand
    back its operand unchanged; it reports the row length in limbs (72 for a 2048-bit modulus in the (4,18) configuration)."""
    rng = random.Random(77)
    n = rng.getrandbits(2048) | (1 << 2047) | 1
    mod = engine.modulus(n)
    xs = [0, 1, n - 1] + [rng.randrange(n) for _ in range(30)]
    out, row_limbs = engine.table_traffic_probe(mod, engine.upload(xs, mod.nwords), 8, 40)
    assert engine.download(out) == xs and row_limbs == 72
    with pytest.raises(ValueError):
        engine.table_traffic_probe(mod, engine.upload(xs, mod.nwords), 0, 40)


def test_c_abi_without_torch_buffers():
    """The boundary is a plain C ABI: drive it with ctypes only (sc_malloc / sc_memcpy_*), no torch tensor in sight."""
    import ctypes as C

    import numpy as np

    from protocols.secure_comparison_amd import _lib
    from protocols.secure_comparison_amd.limbs import ints_to_words, words_to_ints

    lib = _lib.load()
    ctx = C.c_void_p()
    assert lib.sc_ctx_create(0, C.byref(ctx)) == 0
    try:
        rng = random.Random(77)
        n = rng.getrandbits(2048) | (1 << 2047) | 1
        nw, count = 64, 33
        a, b = [rng.randrange(n) for _ in range(count)], [rng.randrange(n) for _ in range(count)]
        mod, exp = C.c_int(), C.c_int()
        n_words = ints_to_words([n], nw)
        assert lib.sc_mod_create(ctx, n_words.ctypes.data_as(C.c_void_p), nw, C.byref(mod)) == 0
        assert lib.sc_mod_words(ctx, mod.value) == nw
        e = 0x10001
        e_words = ints_to_words([e], 1)
        assert lib.sc_exp_create(ctx, e_words.ctypes.data_as(C.c_void_p), 1, C.byref(exp)) == 0
        bufs = []
        for _ in range(3):
            p = C.c_void_p()
            assert lib.sc_malloc(ctx, count * nw * 4, C.byref(p)) == 0
            bufs.append(p)
        for p, vals in zip(bufs, (a, b)):
            host = ints_to_words(vals, nw)
            assert lib.sc_memcpy_h2d(ctx, p, host.ctypes.data_as(C.c_void_p), host.nbytes) == 0
        out = np.zeros((count, nw), dtype=np.uint32)
        assert lib.sc_modmul(ctx, mod.value, bufs[0], nw, bufs[1], nw, bufs[2], count) == 0
        assert lib.sc_memcpy_d2h(ctx, out.ctypes.data_as(C.c_void_p), bufs[2], out.nbytes) == 0
        assert words_to_ints(out) == [x * y % n for x, y in zip(a, b)]
        assert lib.sc_modexp_shared(ctx, mod.value, exp.value, bufs[0], nw, bufs[1], bufs[2], count) == 0
        assert lib.sc_memcpy_d2h(ctx, out.ctypes.data_as(C.c_void_p), bufs[2], out.nbytes) == 0
        assert words_to_ints(out) == [pow(x, e, n) * y % n for x, y in zip(a, b)]
        assert lib.sc_modmul(ctx, mod.value, None, nw, bufs[1], nw, bufs[2], count) == -1      # SC_ERR_ARG, message available
        assert b"sc_modmul" in lib.sc_last_error(ctx)
        for p in bufs:
            assert lib.sc_free(ctx, p) == 0
    finally:
        lib.sc_ctx_destroy(ctx)


@pytest.mark.parametrize("l", [1, 2, 5, 33])
def test_small_and_odd_bit_lengths(engine, keys, l):
    """Edge bit lengths (l = 1: a single bit, no suffix product; l = 33: bit index crosses a 32-bit word) with a DGK key
    generated for the matching u = next_prime(2^(l+2)), every comparison in range enumerated for tiny l."""
    from protocols.secure_comparison_amd.batch import secure_comparison_batch

    sk = oracle_paillier(keys, 1024)
    rng = random.Random(400 + l)
    dgk = o.DGKKey.generate(40, 512, o.next_prime(1 << (l + 2)), rng)
    alice_p, alice_d, bob_p, bob_d = _schemes(engine, sk, dgk, 100)
    pairs = [(x, y) for x in range(1 << l) for y in range(1 << l)] if l <= 2 else \
        [(rng.randrange(1 << l), rng.randrange(1 << l)) for _ in range(12)] + [(5, 5), ((1 << l) - 1, (1 << l) - 1), (0, (1 << l) - 1)]
    drs = [o.draw(rng, l, sk, dgk, 100) for _ in pairs]
    x_enc = [sk.enc_raw(x) for x, _ in pairs]
    y_enc = [sk.enc_raw(y) for _, y in pairs]
    expect = [o.compare(a, b, l, sk, dgk, d, True) for a, b, d in zip(x_enc, y_enc, drs)]
    nw = bob_p.mod_n.nwords
    draws = _draw_tensors(engine, drs, l, nw, (dgk.u.bit_length() + 31) // 32, 4, engine.device)
    got = engine.download(secure_comparison_batch(engine.upload(x_enc, 2 * nw), engine.upload(y_enc, 2 * nw), l, alice_p, alice_d,
                                                  bob_p, bob_d, draws))
    assert got == expect and [sk.dec_raw(c) for c in got] == [int(x <= y) for x, y in pairs]


def test_config3_full_size_properties(engine, keys):
    """BASELINE configs[2] at full size (B = 65536, l = 32, 2048-bit keys), checked through size-independent properties:
    Dec(result) == [x <= y] for every comparison; the randomized run and the static (unrandomized) run decrypt alike but differ
    as ciphertexts; re-randomizing a ciphertext keeps its plaintext; Dec(Enc(m)) round-trips for 65536 plaintexts."""
    import bench
    from protocols.secure_comparison_amd.batch import secure_comparison_batch

    sk, dgk = oracle_paillier(keys, 2048), oracle_dgk(keys, "dgk_2048_l32")
    alice_p, alice_d, bob_p, bob_d = _schemes(engine, sk, dgk, 400)
    B, l = 65536, 32
    x, y, x_enc, y_enc, draws = bench.synth_inputs(engine, l, alice_p, bob_p, bob_d, B, 400, seed=5)
    expect = (x <= y).to(torch.int32)
    res = secure_comparison_batch(x_enc, y_enc, l, alice_p, alice_d, bob_p, bob_d, draws, randomize=True)
    dec = bob_p.decrypt_raw_batch(res)
    assert bool(((dec[:, 0] == expect) & (dec[:, 1:] == 0).all(dim=1)).all().item())
    static = secure_comparison_batch(x_enc, y_enc, l, alice_p, alice_d, bob_p, bob_d, draws, randomize=False)
    assert bool((bob_p.decrypt_raw_batch(static)[:, 0] == expect).all().item())
    assert not bool((static == res).all(dim=1).any().item())              # every ciphertext differs once randomized
    rer = alice_p.randomize_batch(res, draws.rho_zeta_1)
    assert bool((bob_p.decrypt_raw_batch(rer)[:, 0] == expect).all().item()) and not bool((rer == res).all(dim=1).any().item())
    m = draws.r                                                            # 65536 plaintexts below N
    assert bool((bob_p.decrypt_raw_batch(alice_p.randomize_batch(alice_p.encrypt_raw_batch(m), draws.rho_z)) == m).all().item())


@pytest.mark.parametrize("bits", [256, 512, 1024, 1536, 2048, 3072])
def test_pair_arithmetic_modexp(engine, bits):
    """sc_modexp_shared_sq (x^e mod m^2 with Montgomery products modulo m only) against Python pow: operands of 1, 2 and 4
    chunks (wider than m^2 included), edge operands 0, 1, m, m - 1, m^2 - 1, exponents 1, 2, 3, m, m - 1 and random, mul_into.
    1536 and 3072 bits run the pair kernel in an internal twin context (L = 14) of a modulus whose own configuration is L = 27."""
    rng = random.Random(bits)
    m = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    mm, mm2 = engine.modulus(m), engine.modulus(m * m, 2 * ((bits + 31) // 32))
    assert engine.supports_sq(mm)
    B = 24 if bits < 3072 else 12              # (the L = 14 twin of 3072 bits holds 8 numbers per wave: still more than one wave)
    for mult, xs in ((1, [rng.randrange(m) for _ in range(B)]), (2, [rng.randrange(m * m) for _ in range(B)]), (4, [rng.getrandbits(4 * bits) for _ in range(B)])):
        xs[:5] = [1, 0, m - 1, m if mult > 1 else 2, (m * m - 1) if mult > 1 else m - 2]
        t = engine.upload(xs, mult * mm.nwords)
        for e in (1, 2, 3, m, m - 1, rng.getrandbits(bits) | 1, (1 << 77) - 1):
            assert engine.download(engine.modexp_shared_sq(mm, mm2, t, e)) == [pow(x, e, m * m) for x in xs], (bits, mult, e.bit_length())
        cs = [rng.randrange(m * m) for _ in range(B)]
        got = engine.download(engine.modexp_shared_sq(mm, mm2, t, m, mul_into=engine.upload(cs, mm2.nwords)))
        assert got == [pow(x, m, m * m) * c % (m * m) for x, c in zip(xs, cs)]
    n = (1 << bits) - 1                      # all-ones limbs: the largest column sums of the pair passes
    while n % 3 == 0 or n % 5 == 0:
        n -= 2
    mn, mn2 = engine.modulus(n), engine.modulus(n * n, 2 * ((bits + 31) // 32))
    xs = [n * n - 1, n - 1, n * n - n, (1 << (2 * bits - 1)) - 1]
    assert engine.download(engine.modexp_shared_sq(mn, mn2, engine.upload(xs, 2 * mn.nwords), (1 << 100) - 1)) == [pow(x, (1 << 100) - 1, n * n) for x in xs]
    with pytest.raises(Exception):
        engine.modexp_shared_sq(mm, mn2, t, 3)   # mod_m2 is not the square of mod_m


def test_paillier_paths_agree(engine, keys):
    """Pair arithmetic on/off and CRT on/off give identical ciphertexts and plaintexts (2048-bit key)."""
    from protocols.secure_comparison_amd import Paillier

    sk = oracle_paillier(keys, 2048)
    rng = random.Random(4)
    B = 30
    ms = [rng.randrange(sk.n) for _ in range(B)]
    rhos = [1 + rng.randrange(sk.n - 1) for _ in range(B)]
    outs = []
    for use_crt in (False, True):
        for use_pairs in (False, True):
            p = Paillier(sk.n, sk.p, sk.q, engine=engine, use_crt=use_crt, use_pairs=use_pairs)
            nw = p.mod_n.nwords
            rnd = p.randomize_batch(p.encrypt_raw_batch(engine.upload(ms, nw)), engine.upload(rhos, nw))
            outs.append((engine.download(rnd), engine.download(p.decrypt_raw_batch(rnd))))
    assert all(o_ == outs[0] for o_ in outs) and outs[0][1] == ms
    assert outs[0][0] == [sk.randomize(sk.enc_raw(m), r) for m, r in zip(ms, rhos)]


@pytest.mark.parametrize("bits", [512, 1024, 2048, 4000])
def test_latency_configurations_primitives(latency_engine, bits):
    """The small-batch kernels ((2G, 9) lanes x limbs on the limb arrays of an L = 18 modulus) give the same residues as Python
    ints: products, shared and per-element exponentiation, inversion, wide operands, all-ones limbs, pair arithmetic."""
    engine = latency_engine
    rng = random.Random(9000 + bits)
    for n in (rng.getrandbits(bits) | (1 << (bits - 1)) | 1, (1 << bits) - 1 - 2 * 44):
        while n % 3 == 0 or n % 5 == 0:
            n -= 2
        mod = engine.modulus(n)
        vals = [n - 1, n - 2, 1, 0, (1 << (bits - 1)) - 1] + [rng.randrange(n) for _ in range(28)]
        t = engine.upload(vals, mod.nwords)
        e = rng.getrandbits(300) | 1
        assert engine.download(engine.modmul(mod, t, t)) == [v * v % n for v in vals]
        assert engine.download(engine.modexp_shared(mod, t, e, mul_into=t)) == [pow(v, e, n) * v % n for v in vals]
        ev = [rng.getrandbits(67) for _ in vals]
        assert engine.download(engine.modexp_var(mod, t, engine.upload(ev, 3), 67)) == [pow(v, x, n) for v, x in zip(vals, ev)]
        wide = [rng.getrandbits(2 * bits) for _ in range(9)]
        assert engine.download(engine.modexp_shared(mod, engine.upload(wide, 2 * mod.nwords), 3)) == [pow(w, 3, n) for w in wide]
        import math
        inv_in = [v for v in vals if v and math.gcd(v, n) == 1] * 3       # 90+ elements: tree levels above the xgcd top
        assert engine.download(engine.modinv(mod, engine.upload(inv_in, mod.nwords))) == [pow(v, -1, n) for v in inv_in]
        if bits <= 2048:
            mod2 = engine.modulus(n * n, 2 * mod.nwords)
            assert engine.supports_sq(mod)
            got = engine.download(engine.modexp_shared_sq(mod, mod2, t, n, mul_into=engine.upload([3] * len(vals), mod2.nwords)))
            assert got == [pow(v, n, n * n) * 3 % (n * n) for v in vals]


def test_latency_configurations_whole_comparison(latency_engine, keys):
    """A batch of whole comparisons (2048-bit keys, l = 32, CRT key holder) through the small-batch kernels: bit-exact vs the oracle."""
    engine = latency_engine
    sk, dgk = oracle_paillier(keys, 2048), oracle_dgk(keys, "dgk_2048_l32")
    l, B, rbits = 32, 24, 400
    alice_p, alice_d, bob_p, bob_d = _schemes(engine, sk, dgk, rbits)
    rng = random.Random(4242)
    xs = [rng.randrange(1 << l) for _ in range(B)]
    ys = [x if i % 4 == 0 else rng.randrange(1 << l) for i, x in enumerate(xs)]
    drs = [o.draw(rng, l, sk, dgk, rbits) for _ in range(B)]
    nw = alice_p.mod_n.nwords
    x_enc = [sk.randomize(sk.enc_raw(x), 1 + rng.randrange(sk.n - 1)) for x in xs]
    y_enc = [sk.randomize(sk.enc_raw(y), 1 + rng.randrange(sk.n - 1)) for y in ys]
    draws = _draw_tensors(engine, drs, l, nw, (dgk.u.bit_length() + 31) // 32, (rbits + 31) // 32, engine.device)
    from protocols.secure_comparison_amd.batch import secure_comparison_batch

    res = secure_comparison_batch(engine.upload(x_enc, 2 * nw), engine.upload(y_enc, 2 * nw), l, alice_p, alice_d, bob_p, bob_d, draws, randomize=True)
    expect = [o.compare(a, b, l, sk, dgk, dr, randomize=True) for a, b, dr in zip(x_enc, y_enc, drs)]
    assert engine.download(res) == expect
    assert [sk.dec_raw(c) for c in expect] == [int(x <= y) for x, y in zip(xs, ys)]


def test_fresh_context_on_a_side_stream(keys):
    """Everything created lazily (modulus contexts, constants, programs, fixed-base and CRT tables) while the caller's current
    stream is a non-blocking side stream: set-up copies must be ordered on that stream, not on the null stream (regression:
    a program's constants were copied on the null stream and could land after the first launch).  Two contexts on two streams
    driven by two host threads, each bit-exact against the oracle."""
    import threading

    from protocols.secure_comparison_amd import DGK, Paillier
    from protocols.secure_comparison_amd.batch import secure_comparison_batch
    from protocols.secure_comparison_amd.engine import Engine

    sk, dgk = oracle_paillier(keys, 1024), oracle_dgk(keys, "dgk_1024_l16")
    l, B, rbits = 16, 48, 400
    results, errors = {}, []

    def worker(k):
        try:
            rng = random.Random(77 + k)
            with torch.cuda.stream(torch.cuda.Stream()):
                eng = Engine()
                bob_p = Paillier(sk.n, sk.p, sk.q, engine=eng)
                bob_d = DGK(dgk.n, dgk.g, dgk.h, dgk.u, dgk.t, dgk.p, dgk.q, dgk.v_p, dgk.v_q, engine=eng, randomizer_bits=rbits)
                alice_p, alice_d = bob_p.public_copy(), bob_d.public_copy()
                xs = [rng.randrange(1 << l) for _ in range(B)]
                ys = [rng.randrange(1 << l) for _ in range(B)]
                drs = [o.draw(rng, l, sk, dgk, rbits) for _ in range(B)]
                x_enc = [sk.randomize(sk.enc_raw(x), 1 + rng.randrange(sk.n - 1)) for x in xs]
                y_enc = [sk.randomize(sk.enc_raw(y), 1 + rng.randrange(sk.n - 1)) for y in ys]
                nw = bob_p.mod_n.nwords
                draws = _draw_tensors(eng, drs, l, nw, (dgk.u.bit_length() + 31) // 32, (rbits + 31) // 32, eng.device)
                got = eng.download(secure_comparison_batch(eng.upload(x_enc, 2 * nw), eng.upload(y_enc, 2 * nw), l, alice_p, alice_d,
                                                           bob_p, bob_d, draws))
                results[k] = (got, [o.compare(a, b, l, sk, dgk, d, True) for a, b, d in zip(x_enc, y_enc, drs)])
        except Exception as exc:  # surfaced in the main thread below
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    for k in range(2):
        assert results[k][0] == results[k][1]


def test_concurrent_shards_equal_single_stream(engine, keys):
    """batch.ConcurrentShards (two library contexts, two HIP streams, two host threads) returns the very residues of the
    single-stream batch call, shard by shard, incl. a ragged split and a shuffle."""
    from protocols.secure_comparison_amd import DGK, Paillier
    from protocols.secure_comparison_amd.batch import ConcurrentShards, PartySet, secure_comparison_batch, split_draws
    from protocols.secure_comparison_amd.distributed import shard_bounds
    from protocols.secure_comparison_amd.engine import Engine

    sk, dgk = oracle_paillier(keys, 1024), oracle_dgk(keys, "dgk_1024_l16")
    l, B, rbits = 16, 37, 400
    rng = random.Random(99)
    alice_p, alice_d, bob_p, bob_d = _schemes(engine, sk, dgk, rbits)
    xs = [rng.randrange(1 << l) for _ in range(B)]
    ys = [rng.randrange(1 << l) for _ in range(B)]
    drs = [o.draw(rng, l, sk, dgk, rbits) for _ in range(B)]
    nw = bob_p.mod_n.nwords
    x_enc = engine.upload([sk.randomize(sk.enc_raw(x), 1 + rng.randrange(sk.n - 1)) for x in xs], 2 * nw)
    y_enc = engine.upload([sk.randomize(sk.enc_raw(y), 1 + rng.randrange(sk.n - 1)) for y in ys], 2 * nw)
    draws = _draw_tensors(engine, drs, l, nw, (dgk.u.bit_length() + 31) // 32, (rbits + 31) // 32, engine.device)
    single = secure_comparison_batch(x_enc, y_enc, l, alice_p, alice_d, bob_p, bob_d, draws)

    parties = [PartySet(alice_p, alice_d, bob_p, bob_d, torch.cuda.Stream())]
    e2 = Engine()
    bp2 = Paillier(sk.n, sk.p, sk.q, engine=e2)
    bd2 = DGK(dgk.n, dgk.g, dgk.h, dgk.u, dgk.t, dgk.p, dgk.q, dgk.v_p, dgk.v_q, engine=e2, randomizer_bits=rbits)
    parties.append(PartySet(bp2.public_copy(), bd2.public_copy(), bp2, bd2, torch.cuda.Stream()))
    bounds = [shard_bounds(B, i, 2) for i in range(2)]
    shards = [(x_enc[a:b].contiguous(), y_enc[a:b].contiguous(), d) for (a, b), d in zip(bounds, split_draws(draws, bounds))]
    runner = ConcurrentShards(parties)
    try:
        for _ in range(3):
            parts = runner.run(shards, l)
            assert torch.equal(torch.cat(parts, dim=0), single)
    finally:
        runner.close()
    assert [sk.dec_raw(c) for c in engine.download(single)] == [int(x <= y) for x, y in zip(xs, ys)]
    with pytest.raises(ValueError):
        ConcurrentShards([parties[0], parties[0]])
