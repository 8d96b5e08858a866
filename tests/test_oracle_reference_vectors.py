"""Pins the CPU oracle at plaintext level with the REFERENCE's own test vectors and per-step identities
(/root/reference/src/tno/mpc/protocols/secure_comparison/test/unit/test_secure_comparison.py:27-68 vectors,
:156-800 identities; scheme parameters as at :19-24: Paillier 1024-bit, DGK v_bits=20 n_bits=128 u=next_prime(2^18),
l=16).  The reference has no golden ciphertexts, so these identities are all it pins."""
import random

import pytest

from conftest import oracle_dgk, oracle_paillier
from oracle import sc_oracle as o

L = 16
PAIRS = [(-400, -383), (-1, 0), (0, 2), (1, 10), (230, 269), (1508, 2408), (3122, 6048), (4250, 7804), (8668, 9015)]


@pytest.fixture(scope="module")
def schemes(keys):
    return oracle_paillier(keys, 1024), oracle_dgk(keys, "dgk_tiny_l16")


def chain(sk, dgk, x, y, rng, upto):
    """The static step chain of run_comparison (:71-124), stopping after step `upto`."""
    st = {}
    st["x_enc"], st["y_enc"] = sk.enc_raw(sk.encode(x)), sk.enc_raw(sk.encode(y))
    st["z_enc"], st["r"] = o.step_1(st["x_enc"], st["y_enc"], L, sk, rng.randrange(sk.n))
    st["z"], st["beta"] = o.step_2(st["z_enc"], L, sk)
    st["alpha"] = o.step_3(st["r"], L)
    st["d_enc"] = o.step_4c(o.step_4a(st["z"], dgk, sk, L), st["r"], dgk, sk)
    st["beta_enc"] = o.step_4b(st["beta"], L, dgk)
    st["xor"] = o.step_4d(st["alpha"], st["beta_enc"], dgk)
    st["w_e"], st["alpha_tilde"] = o.step_4e(st["r"], st["alpha"], st["xor"], st["d_enc"], sk, dgk)
    st["w"] = o.step_4f(st["w_e"], dgk)
    st["s"], st["delta_a"] = o.step_4g(rng.randrange(2))
    st["c"] = o.step_4h(st["s"], st["alpha"], st["alpha_tilde"], st["d_enc"], st["beta_enc"], st["w"], st["delta_a"], dgk)
    return st


@pytest.mark.parametrize("number", [v for pair in PAIRS for v in pair if v >= 0])
def test_bit_conversion(number):  # :156-163
    assert o.from_bits(o.to_bits(number, L)) == number


def test_modulo_n_squared(schemes):  # :166-182
    sk, _ = schemes
    x_enc = sk.enc_raw(0)
    y_enc = sk.n2 - 2
    y = sk.dec_raw(y_enc)
    r = sk.n2
    z_enc = sk.add(sk.add(y_enc, sk.neg(x_enc)), sk.enc_raw((1 << L) + r))
    assert (y - 0 + (1 << L) + r) % sk.n2 == sk.dec_raw(z_enc)


@pytest.mark.parametrize("x, y", PAIRS)
def test_step_1(schemes, x, y):  # :196-217
    sk, _ = schemes
    z_enc, r = o.step_1(sk.enc_raw(sk.encode(x)), sk.enc_raw(sk.encode(y)), L, sk, random.Random(x * 7 + y).randrange(sk.n))
    assert sk.decode(sk.dec_raw(sk.enc_raw((y - x + (1 << L) + r) % sk.n2))) == sk.decode(sk.dec_raw(z_enc))


def test_step_1_constraint_rejected(schemes):  # :240-252
    sk, _ = schemes
    with pytest.raises(AssertionError):
        o.step_1(sk.enc_raw(1), sk.enc_raw(2), sk.n.bit_length() - 2, sk, 5)


@pytest.mark.parametrize("x, y", PAIRS)
def test_step_2_and_4b(schemes, x, y):  # :263-277, :303-330
    sk, dgk = schemes
    st = chain(sk, dgk, x, y, random.Random(x + 3 * y), "4b")
    assert st["z"] == (y - x + (1 << L) + st["r"]) % sk.n and st["beta"] == st["z"] % 2 ** L
    bits = [int(b) for b in reversed(bin(st["beta"])[2:])]
    bits += [0] * (L - len(bits))
    assert [dgk.decrypt_full(c) for c in st["beta_enc"]] == bits


def test_step_4a(schemes):  # :280-300, TEST_4A = 0, N, (N-1)//2
    sk, dgk = schemes
    for z in (0, sk.n, (sk.n - 1) // 2):
        assert dgk.decrypt_full(o.step_4a(z, dgk, sk, L)) == int(z < (sk.n - 1) // 2)


def test_step_4c(schemes):  # :333-387
    sk, dgk = schemes
    for d in (0, 1):
        for r in (0, 100):
            assert dgk.decrypt_full(o.step_4c(dgk.enc_raw(d), r, dgk, sk)) == 0
        assert dgk.decrypt_full(o.step_4c(dgk.enc_raw(d), sk.n - 100, dgk, sk)) == d
    for r in (-10, sk.n + 10):
        with pytest.raises(AssertionError):
            o.step_4c(dgk.enc_raw(0), r, dgk, sk)


@pytest.mark.parametrize("a, b", [(12345, 23456), (25, 890)])
def test_step_4d(schemes, a, b):  # :390-407, TEST_4D
    _, dgk = schemes
    alpha, beta = o.to_bits(a, L), o.to_bits(b, L)
    xor = o.step_4d(alpha, [dgk.enc_raw(v) for v in beta], dgk)
    assert [dgk.decrypt_full(c) for c in xor] == [x ^ y for x, y in zip(alpha, beta)]


@pytest.mark.parametrize("x, y", PAIRS)
def test_steps_4e_4f_4h_4j(schemes, x, y):  # :410-545
    sk, dgk = schemes
    rng = random.Random(1000 + x * 31 + y)
    st = chain(sk, dgk, x, y, rng, "4h")
    u = dgk.u
    d = dgk.decrypt_full(st["d_enc"])
    beta_bits = o.to_bits(st["beta"], L)
    assert st["alpha_tilde"] == o.to_bits((st["r"] - sk.n) % (1 << L), L)
    w_plain = []
    for i in range(L):
        expect = (st["alpha"][i] ^ beta_bits[i]) - (0 if st["alpha"][i] == st["alpha_tilde"][i] else d)
        assert dgk.decrypt_full(st["w_e"][i]) == expect % u                       # 4e
        assert dgk.decrypt_full(st["w"][i]) == expect * 2 ** i % u                 # 4f
        w_plain.append(expect * 2 ** i % u)
    for i in range(L):                                                             # 4h (:496-507)
        t = st["s"] + st["alpha"][i] + (st["alpha_tilde"][i] - st["alpha"][i]) * d - beta_bits[i] + 3 * sum(w_plain[i + 1:])
        assert dgk.decrypt_full(st["c"][i + 1]) == t % u
    assert dgk.decrypt_full(st["c"][0]) == (st["delta_a"] + sum(w_plain)) % u
    rhos = [1 + rng.randrange(u - 1) for _ in range(L + 1)]
    blinded = o.step_4i(st["c"], dgk, rhos, None)
    delta_b = o.step_4j(blinded, dgk)                                              # 4j
    assert delta_b == int(any(dgk.decrypt_full(c) == 0 for c in st["c"]))
    assert [dgk.is_zero(c) for c in blinded] == [dgk.decrypt_full(c) == 0 for c in st["c"]]


def test_step_5(schemes):  # :548-566, TEST_5
    sk, _ = schemes
    n = sk.n
    for z, delta_b in ((n, 0), (100, 0), (n, 1), (100, 1)):
        z1, z2, db = o.step_5(z, L, delta_b, sk)
        assert sk.dec_raw(z1) == z // 2 ** L % n
        assert sk.dec_raw(z2) == (int(z < (n - 1) // 2) * ((z + n) // 2 ** L) + int(z >= (n - 1) // 2) * (z // 2 ** L)) % n
        assert sk.dec_raw(db) == delta_b


def test_step_6(schemes):  # :569-586, TEST_6
    sk, _ = schemes
    for delta_a in (0, 1):
        for delta_b in (0, 1):
            got = sk.dec_raw(o.step_6(delta_a, sk.enc_raw(delta_b), sk))
            assert got == (delta_b if delta_a == 1 else 1 - delta_b)


@pytest.mark.parametrize("x, y", PAIRS)
def test_end_to_end_both_modes(schemes, x, y):  # :641-800: smaller / greater / equal, static and interactive flows
    sk, dgk = schemes
    rng = random.Random(x * 131 + y)
    for a, b, expect in ((x, y, 1), (y, x, 0), (x, x, 1), (y, y, 1)):
        for randomize in (False, True):
            dr = o.draw(rng, L, sk, dgk, 50)
            res = o.compare(sk.enc_raw(sk.encode(a)), sk.enc_raw(sk.encode(b)), L, sk, dgk, dr, randomize)
            assert sk.dec_raw(res) == expect
