"""
CPU oracle for the secure-comparison hot path  --  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  Nothing under ``protocols/`` imports it: the product path runs on the HIP library
and fails loudly when that library is missing.

What it is
----------
A plain-Python-``int`` restatement of the arithmetic that sits underneath the reference's
``Initiator.step_*`` / ``KeyHolder.step_*`` (file:line citations are into ``/root/reference``;
``SC/`` = ``src/tno/mpc/protocols/secure_comparison/``):

* the protocol steps follow ``SC/initiator.py:228-564`` and ``SC/keyholder.py:181-287``;
* ``to_bits`` / ``from_bits`` follow ``SC/utils.py:6-38``;
* the Paillier / DGK scheme arithmetic lives in third-party packages that are NOT vendored in
  ``/root/reference`` (``tno.mpc.encryption_schemes.{paillier,dgk,templates,utils}``,
  ``pyproject.toml:32-38``: ``paillier~=3.0``, ``dgk~=3.0``, ``templates~=4.1,>=4.1.3``,
  ``utils~=0.10``; compatible-release ranges, no lock file).  Their published algorithms are
  restated here from first principles: Paillier with g = N+1, DGK with
  ``Enc(m) = g^m h^r mod n`` and zero test ``c^{v_p} mod p == 1``, anchored on the reference's own
  call sites (``SC/initiator.py:96,249-256,287-290,320,371,406,460-484,503-512,531,559-563``;
  ``SC/keyholder.py:156-166,195,212-216,231,249,274-286``).

Pinning status
--------------
* Plaintext level: PINNED by the reference's own test vectors and per-step identities
  (``SC/test/unit/test_secure_comparison.py:27-68`` vectors, ``:156-800`` identities) --
  ``tests/test_oracle_reference_vectors.py`` replays them against this oracle; ``to_bits`` /
  ``from_bits`` are additionally pinned by vectors generated from the reference's own
  ``SC/utils.py`` (``tests/golden/gen_utils_vectors.py``).
* Ciphertext level (bit patterns): **parity unpinned**.  The reference holds no golden
  ciphertexts / KATs, its randomness comes from ``secrets`` (unseedable) and the scheme packages
  cannot be imported here (ordinary ModuleNotFoundError, packages absent).  Every homomorphic
  operation is a deterministic group operation on canonical residues, so any correct
  implementation yields the same integers once the same random inputs are injected; that is the
  contract the GPU path is tested against.

Every random draw of the reference (``r`` SC/initiator.py:250, ``delta_a`` :420, ``rho_i`` :512,
the shuffle :223, and the scheme randomizers) is an explicit argument here.
"""

from __future__ import annotations

import random
from dataclasses import dataclass, field
from typing import Sequence

try:  # optional acceleration, identical results (the reference does the same: README.md:49)
    import gmpy2  # type: ignore

    _HAVE_GMPY2 = True
except Exception:  # pragma: no cover - absent in the default interpreter
    gmpy2 = None
    _HAVE_GMPY2 = False


# --------------------------------------------------------------------------- L0: integers
def pow_mod(base: int, exp: int, mod: int) -> int:
    """base**exp mod mod; negative exponents invert first ([ext] utils.pow_mod semantics)."""
    if _HAVE_GMPY2:
        return int(gmpy2.powmod(base, exp, mod))
    return pow(base, exp, mod)


def mod_inv(x: int, mod: int) -> int:
    """Modular inverse; raises ZeroDivisionError/ValueError when gcd(x, mod) != 1."""
    if _HAVE_GMPY2:
        return int(gmpy2.invert(x, mod))
    return pow(x, -1, mod)


_SMALL_PRIMES = [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97]


def is_probable_prime(n: int, rounds: int = 24, rng: random.Random | None = None) -> bool:
    """Miller-Rabin with fixed small bases plus random bases."""
    if n < 2:
        return False
    for p in _SMALL_PRIMES:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    rng = rng or random.Random(n & 0xFFFFFFFF)
    bases = _SMALL_PRIMES[:12] + [rng.randrange(2, n - 1) for _ in range(rounds)]
    for a in bases:
        a %= n
        if a in (0, 1, n - 1):
            continue
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def next_prime(n: int) -> int:
    """Smallest prime > n ([ext] utils.next_prime, used at SC/keyholder.py:164)."""
    c = n + 1
    if c <= 2:
        return 2
    if c % 2 == 0:
        c += 1
    while not is_probable_prime(c):
        c += 2
    return c


def rand_prime(bits: int, rng: random.Random) -> int:
    """Random prime with exactly `bits` bits (top bit set)."""
    while True:
        c = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        if is_probable_prime(c, rng=rng):
            return c


# --------------------------------------------------------------------------- utils.py
def to_bits(integer: int, bit_length: int) -> list[int]:
    """LSB-first bits (SC/utils.py:6-21); asserts integer < 2**bit_length (:16)."""
    assert integer < (1 << bit_length)
    return [(integer >> i) & 1 for i in range(bit_length)]


def from_bits(bits: Sequence[int]) -> int:
    """Inverse of to_bits (SC/utils.py:24-38): only bits equal to 1 contribute."""
    return sum(1 << i for i, b in enumerate(bits) if b == 1)


# --------------------------------------------------------------------------- Paillier
@dataclass
class PaillierKey:
    """Paillier key with g = N + 1.  p, q may be None (public part only)."""

    n: int
    p: int | None = None
    q: int | None = None
    n2: int = field(init=False)
    lam: int | None = field(init=False, default=None)
    mu: int | None = field(init=False, default=None)

    def __post_init__(self) -> None:
        self.n2 = self.n * self.n
        if self.p is not None and self.q is not None:
            assert self.p * self.q == self.n
            self.lam = (self.p - 1) * (self.q - 1)
            self.mu = mod_inv(self.lam, self.n)

    @classmethod
    def generate(cls, key_length: int, rng: random.Random) -> "PaillierKey":
        while True:
            p = rand_prime(key_length // 2, rng)
            q = rand_prime(key_length - key_length // 2, rng)
            if p != q and (p * q).bit_length() == key_length:
                return cls(p * q, p, q)

    def public(self) -> "PaillierKey":
        return PaillierKey(self.n)

    # [ext] encoding used only at the API edge (SC/initiator.py:93-102, default apply_encoding=True)
    def encode(self, m: int) -> int:
        return m % self.n

    def decode(self, m: int) -> int:
        return m - self.n if m > self.n // 2 else m

    def enc_raw(self, m: int) -> int:
        """Unrandomized encryption (N+1)^m = 1 + (m mod N) N  (mod N^2)."""
        return (1 + (m % self.n) * self.n) % self.n2

    def randomizer(self, rho: int) -> int:
        """rho^N mod N^2 (the value a `.randomize()` multiplies in, SC/initiator.py:109)."""
        return pow_mod(rho, self.n, self.n2)

    def randomize(self, c: int, rho: int) -> int:
        return c * self.randomizer(rho) % self.n2

    def dec_raw(self, c: int) -> int:
        """m = L(c^lam mod N^2) * mu mod N (SC/keyholder.py:195, apply_encoding=False)."""
        assert self.lam is not None and self.mu is not None
        x = pow_mod(c, self.lam, self.n2)
        return (x - 1) // self.n * self.mu % self.n

    # homomorphisms = the ciphertext operator algebra of SURVEY 8(a)/a21
    def add(self, c1: int, c2: int) -> int:
        return c1 * c2 % self.n2

    def neg(self, c: int) -> int:
        return mod_inv(c, self.n2)

    def mul(self, c: int, k: int) -> int:
        return pow_mod(c, k, self.n2)


# --------------------------------------------------------------------------- DGK
@dataclass
class DGKKey:
    """DGK key.  Secret part (p, q, v_p, v_q) may be None."""

    n: int
    g: int
    h: int
    u: int
    t: int
    p: int | None = None
    q: int | None = None
    v_p: int | None = None
    v_q: int | None = None

    @classmethod
    def generate(cls, v_bits: int, n_bits: int, u: int, rng: random.Random) -> "DGKKey":
        """Keys as in SC/keyholder.py:161-166 (v_bits, n_bits, u): p = 2 u v_p p_r + 1, etc."""
        half = n_bits // 2
        v_p = rand_prime(v_bits, rng)
        v_q = rand_prime(v_bits, rng)
        while v_q == v_p:
            v_q = rand_prime(v_bits, rng)

        def make_prime(v: int, bits: int) -> tuple[int, int]:
            base = 2 * u * v
            need = bits - base.bit_length()
            assert need > 8, "n_bits too small for u and v_bits"
            while True:
                pr = rng.getrandbits(need + 1) | 1
                cand = base * pr + 1
                if cand.bit_length() == bits and is_probable_prime(cand, rng=rng):
                    return cand, pr

        while True:
            p, p_r = make_prime(v_p, half)
            q, q_r = make_prime(v_q, n_bits - half)
            if p != q and (p * q).bit_length() == n_bits:
                break
        n = p * q

        def elem_of_order(prime: int, cof: int, order_factors: Sequence[int]) -> int:
            # random x^cof has order dividing prod(order_factors); demand exactly that order
            order = 1
            for f in order_factors:
                order *= f
            while True:
                x = rng.randrange(2, prime - 1)
                e = pow(x, cof, prime)
                if e == 1:
                    continue
                if all(pow(e, order // f, prime) != 1 for f in order_factors):
                    return e

        g_p = elem_of_order(p, 2 * p_r, (u, v_p))
        g_q = elem_of_order(q, 2 * q_r, (u, v_q))
        h_p = elem_of_order(p, 2 * p_r * u, (v_p,))
        h_q = elem_of_order(q, 2 * q_r * u, (v_q,))
        q_inv_p = mod_inv(q, p)

        def crt(a_p: int, a_q: int) -> int:
            return (a_q + q * ((a_p - a_q) * q_inv_p % p)) % n

        return cls(n, crt(g_p, g_q), crt(h_p, h_q), u, v_bits, p, q, v_p, v_q)

    def public(self) -> "DGKKey":
        return DGKKey(self.n, self.g, self.h, self.u, self.t)

    def enc_raw(self, m: int) -> int:
        """Unrandomized g^m mod n; negative m inverts g first (SURVEY 8(a) note 2)."""
        return pow_mod(self.g, m, self.n)

    def randomizer(self, r: int) -> int:
        return pow_mod(self.h, r, self.n)

    def randomize(self, c: int, r: int) -> int:
        return c * self.randomizer(r) % self.n

    def add(self, c1: int, c2: int) -> int:
        return c1 * c2 % self.n

    def neg(self, c: int) -> int:
        return mod_inv(c, self.n)

    def mul(self, c: int, k: int) -> int:
        return pow_mod(c, k, self.n)

    def is_zero(self, c: int) -> bool:
        """[ext] DGK.is_zero: c^{v_p} mod p == 1 (SC/keyholder.py:249)."""
        assert self.p is not None and self.v_p is not None
        return pow_mod(c % self.p, self.v_p, self.p) == 1

    def decrypt_full(self, c: int) -> int:
        """Brute-force full decryption for small u (tests only; `full_decryption=True`)."""
        assert self.p is not None and self.v_p is not None
        target = pow_mod(c % self.p, self.v_p, self.p)
        gv = pow_mod(self.g % self.p, self.v_p, self.p)
        acc = 1
        for m in range(self.u):
            if acc == target:
                return m
            acc = acc * gv % self.p
        raise ValueError("not a valid DGK ciphertext")


# --------------------------------------------------------------------------- protocol steps
def step_1(x_enc: int, y_enc: int, l: int, pk: PaillierKey, r: int) -> tuple[int, int]:
    """[[z]] = [[y]] [[x]]^-1 [[2^l + r]] mod N^2 (SC/initiator.py:228-258); r injected (:250)."""
    assert (1 << (l + 2)) < pk.n // 2  # SC/initiator.py:249
    z_enc = pk.add(pk.add(y_enc, pk.neg(x_enc)), pk.enc_raw((1 << l) + r))
    return z_enc, r


def step_2(z_enc: int, l: int, sk: PaillierKey) -> tuple[int, int]:
    """z = Dec([[z]]), beta = z mod 2^l (SC/keyholder.py:181-196)."""
    z = sk.dec_raw(z_enc)
    return z, z % (1 << l)


def step_3(r: int, l: int) -> list[int]:
    """alpha = bits of r mod 2^l (SC/initiator.py:260-270)."""
    return to_bits(r % (1 << l), l)


def step_4a(z: int, dgk: DGKKey, pk: PaillierKey, l: int) -> int:
    """[d] = Enc_DGK(z < (N-1)//2) (SC/keyholder.py:198-216)."""
    assert dgk.u > (1 << (l + 2))  # SC/keyholder.py:212
    return dgk.enc_raw(int(z < (pk.n - 1) // 2))


def step_4b(beta: int, l: int, dgk: DGKKey) -> list[int]:
    """[beta_i] for the l LSB-first bits of beta (SC/keyholder.py:218-233)."""
    return [dgk.enc_raw(b) for b in to_bits(beta, l)]


def step_4c(d_enc: int, r: int, dgk: DGKKey, pk: PaillierKey) -> int:
    """[d] <- [0] when r < (N-1)//2 (SC/initiator.py:272-291)."""
    assert 0 <= r < pk.n  # SC/initiator.py:286-288
    return dgk.enc_raw(0) if r < (pk.n - 1) // 2 else d_enc


def step_4d(alpha: Sequence[int], beta_is_enc: Sequence[int], dgk: DGKKey) -> list[int]:
    """[alpha_i xor beta_i]: beta_i if alpha_i == 0 else [1] [beta_i]^-1 (SC/initiator.py:293-328)."""
    out = []
    for a_i, b_enc in zip(alpha, beta_is_enc):
        out.append(b_enc if a_i == 0 else dgk.add(dgk.neg(b_enc), dgk.enc_raw(1)))
    return out


def step_4e(r: int, alpha: Sequence[int], xor_enc: Sequence[int], d_enc: int, pk: PaillierKey,
            dgk: DGKKey) -> tuple[list[int], list[int]]:
    """alpha~ = (r - N) mod 2^l; [w_i] = xor_i or xor_i [d]^-1 (SC/initiator.py:330-384)."""
    l = len(xor_enc)
    alpha_tilde = to_bits((r - pk.n) % (1 << l), l)
    w = []
    for a_i, at_i, x_enc in zip(alpha, alpha_tilde, xor_enc):
        w.append(x_enc if a_i == at_i else dgk.add(x_enc, dgk.neg(d_enc)))
    return w, alpha_tilde


def step_4f(w_is_enc: Sequence[int], dgk: DGKKey) -> list[int]:
    """[w_i] <- [w_i]^(2^i) (SC/initiator.py:386-410)."""
    return [dgk.mul(w, 1 << i) for i, w in enumerate(w_is_enc)]


def step_4g(delta_a: int) -> tuple[int, int]:
    """s = 1 - 2 delta_a (SC/initiator.py:412-421); delta_a injected (:420)."""
    return 1 - 2 * delta_a, delta_a


def step_4h(s: int, alpha: Sequence[int], alpha_tilde: Sequence[int], d_enc: int,
            beta_is_enc: Sequence[int], w_is_enc: Sequence[int], delta_a: int, dgk: DGKKey) -> list[int]:
    """[c_i] for i = -1, 0, ..., l-1 in that order (SC/initiator.py:423-485)."""
    l = len(beta_is_enc)
    c = [dgk.enc_raw(s) for _ in range(l)]  # :459-461 (s = -1 -> g^-1)
    d_tab = {-1: dgk.mul(d_enc, -1), 0: dgk.mul(d_enc, 0), 1: dgk.mul(d_enc, 1)}  # :465-469
    w_sum = None  # the int 0 of :462
    for i in range(l - 1, -1, -1):
        term = dgk.add(d_tab[alpha_tilde[i] - alpha[i]], dgk.enc_raw(int(alpha[i])))
        term = dgk.add(term, dgk.neg(beta_is_enc[i]))
        three_w = dgk.enc_raw(0) if w_sum is None else dgk.mul(w_sum, 3)  # `3 * 0` is the int 0
        term = dgk.add(term, three_w)
        c[i] = dgk.add(c[i], term)
        w_sum = dgk.add(dgk.enc_raw(0), w_is_enc[i]) if w_sum is None else dgk.add(w_sum, w_is_enc[i])
    c_m1 = dgk.add(w_sum, dgk.enc_raw(delta_a)) if w_sum is not None else dgk.enc_raw(delta_a)
    return [c_m1] + c  # :484


def step_4i(c_is_enc: Sequence[int], dgk: DGKKey, rhos: Sequence[int],
            perm: Sequence[int] | None = None) -> list[int]:
    """Blind with rho_i in [1, u) (SC/initiator.py:487-516); out[k] = blinded[perm[k]]."""
    assert len(rhos) == len(c_is_enc) and all(1 <= x < dgk.u for x in rhos)
    masked = [dgk.mul(c, rho) for c, rho in zip(c_is_enc, rhos)]
    return masked if perm is None else [masked[k] for k in perm]


def step_4j(c_is_enc: Sequence[int], dgk: DGKKey) -> int:
    """delta_B = OR_i is_zero([c_i]) (SC/keyholder.py:235-253)."""
    return int(any(dgk.is_zero(c) for c in c_is_enc))


def step_5(z: int, l: int, delta_b: int, pk: PaillierKey) -> tuple[int, int, int]:
    """[[zeta_1]], [[zeta_2]], [[delta_B]] (SC/keyholder.py:255-287)."""
    zeta_1 = z >> l
    zeta_2 = (z + pk.n) >> l if z < (pk.n - 1) // 2 else z >> l
    return pk.enc_raw(zeta_1), pk.enc_raw(zeta_2), pk.enc_raw(delta_b)


def step_6(delta_a: int, delta_b_enc: int, pk: PaillierKey) -> int:
    """[[beta < alpha]] (SC/initiator.py:518-531)."""
    return delta_b_enc if delta_a == 1 else pk.add(pk.neg(delta_b_enc), pk.enc_raw(1))


def step_7(zeta_1_enc: int, zeta_2_enc: int, r: int, l: int, beta_lt_alpha_enc: int, pk: PaillierKey) -> int:
    """[[x <= y]] = [[zeta]] ([[r div 2^l]] [[beta<alpha]])^-1 (SC/initiator.py:533-564)."""
    zeta_enc = zeta_1_enc if r < (pk.n - 1) // 2 else zeta_2_enc
    return pk.add(zeta_enc, pk.neg(pk.add(pk.enc_raw(r >> l), beta_lt_alpha_enc)))


# --------------------------------------------------------------------------- whole comparison
@dataclass
class Draws:
    """All random inputs of one comparison (SURVEY 8(a) note 1)."""

    r: int
    delta_a: int
    rhos: list[int]            # l+1 blinding exponents in [1, u)
    perm: list[int] | None     # permutation of range(l+1) or None (do_shuffle=False)
    rho_z: int                 # Paillier randomizer base for [[z]]           (SC/initiator.py:109)
    r_d: int                   # DGK randomizer exponent for [d]              (SC/keyholder.py:106)
    r_beta: list[int]          # l DGK randomizer exponents for [beta_i]      (SC/keyholder.py:107-108)
    r_c: list[int]             # l+1 DGK randomizer exponents for [c_i]       (SC/initiator.py:153-154)
    rho_zeta1: int             # Paillier randomizer bases                    (SC/keyholder.py:126-128)
    rho_zeta2: int
    rho_delta_b: int


def draw(rng: random.Random, l: int, pk: PaillierKey, dgk: DGKKey, rbits: int, shuffle: bool = True) -> Draws:
    perm = list(range(l + 1))
    if shuffle:
        rng.shuffle(perm)
    return Draws(
        r=rng.randrange(pk.n), delta_a=rng.randrange(2),
        rhos=[1 + rng.randrange(dgk.u - 1) for _ in range(l + 1)],
        perm=perm if shuffle else None,
        rho_z=1 + rng.randrange(pk.n - 1), r_d=rng.getrandbits(rbits),
        r_beta=[rng.getrandbits(rbits) for _ in range(l)], r_c=[rng.getrandbits(rbits) for _ in range(l + 1)],
        rho_zeta1=1 + rng.randrange(pk.n - 1), rho_zeta2=1 + rng.randrange(pk.n - 1),
        rho_delta_b=1 + rng.randrange(pk.n - 1),
    )


def compare(x_enc: int, y_enc: int, l: int, sk: PaillierKey, dgk: DGKKey, dr: Draws,
            randomize: bool = True, trace: dict | None = None) -> int:
    """
    One full comparison: the interactive flow of SC/initiator.py:69-175 + SC/keyholder.py:70-133
    (with randomize=True: every `.randomize()` call of those drivers), or the static step chain of
    README.md:117-141 (randomize=False).  Returns [[x <= y]]; `trace` collects intermediates.
    """
    pk = sk
    z_enc, r = step_1(x_enc, y_enc, l, pk, dr.r)
    if randomize:
        z_enc = pk.randomize(z_enc, dr.rho_z)
    z, beta = step_2(z_enc, l, sk)
    alpha = step_3(r, l)
    d_enc = step_4a(z, dgk, pk, l)
    beta_enc = step_4b(beta, l, dgk)
    if randomize:
        d_enc = dgk.randomize(d_enc, dr.r_d)
        beta_enc = [dgk.randomize(b, rr) for b, rr in zip(beta_enc, dr.r_beta)]
    d_sent = d_enc
    d_enc = step_4c(d_enc, r, dgk, pk)
    xor_enc = step_4d(alpha, beta_enc, dgk)
    w_enc, alpha_tilde = step_4e(r, alpha, xor_enc, d_enc, pk, dgk)
    w_enc = step_4f(w_enc, dgk)
    s, delta_a = step_4g(dr.delta_a)
    c_enc_h = step_4h(s, alpha, alpha_tilde, d_enc, beta_enc, w_enc, delta_a, dgk)
    c_enc = step_4i(c_enc_h, dgk, dr.rhos, dr.perm)
    if randomize:
        c_enc = [dgk.randomize(c, rr) for c, rr in zip(c_enc, dr.r_c)]
    delta_b = step_4j(c_enc, dgk)
    z1, z2, db = step_5(z, l, delta_b, pk)
    if randomize:
        z1, z2, db = pk.randomize(z1, dr.rho_zeta1), pk.randomize(z2, dr.rho_zeta2), pk.randomize(db, dr.rho_delta_b)
    blta = step_6(delta_a, db, pk)
    res = step_7(z1, z2, r, l, blta, pk)
    if trace is not None:
        trace.update(z_enc=z_enc, z=z, beta=beta, d_sent=d_sent, beta_enc=beta_enc, c_h=c_enc_h, c_enc=c_enc,
                     delta_b=delta_b, zeta1=z1, zeta2=z2, delta_b_enc=db, result=res)
    return res


# --------------------------------------------------------------------------- limb packing helpers
def int_to_words(x: int, nwords: int) -> list[int]:
    assert 0 <= x < (1 << (32 * nwords))
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(nwords)]


def words_to_int(words: Sequence[int]) -> int:
    return sum(int(w) << (32 * i) for i, w in enumerate(words))
