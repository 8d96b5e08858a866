"""Paillier and DGK scheme objects on top of the HIP engine.

The reference delegates all arithmetic to un-vendored packages (tno.mpc.encryption_schemes.paillier / .dgk /
.templates, pyproject.toml:32-38) and reaches them through the object API listed in SURVEY.md 8(b); the classes
here offer that same surface (method names, argument order, operator algebra, warnings) and add batched twins
that keep whole batches as device arrays.  Every ciphertext operation -- also for a single ciphertext -- runs
on the GPU through libsc_amd.so; there is no CPU arithmetic path (key generation and the test-only DGK
lookup-table decryption are host-side set-up).

Operator algebra:  ct + ct = modular product; ct + int = multiply by Enc(int);
-ct / ct * -1 = modular inverse; ct * k = ct^k (k < 0 inverts first); a - b = a + (b * -1); randomize() multiplies
by rho^N (Paillier) or h^r (DGK).
"""
from __future__ import annotations

import secrets
import warnings
from typing import Any

import torch

from . import keygen
from .limbs import RowBlock

WARN_INEFFICIENT_RANDOMIZATION = (
    "Randomizing a fresh ciphertext wastes randomness: the ciphertext was already randomized and unused."
)
WARN_INEFFICIENT_HOM_OPERATION = (
    "A fresh ciphertext was used as input to a homomorphic operation and is no longer fresh afterwards: its randomness "
    "is wasted if the result is randomized again.  Randomize ciphertexts as late as possible (just before sending)."
)
WARN_UNFRESH_SERIALIZATION = (
    "Serializing a ciphertext that is not fresh: it is randomized first.  Call ciphertext.randomize() before sending."
)
WARN_OUT_OF_RANDOMNESS = (
    "No pre-generated randomness available; generating randomness on the fly "
    "(boot_randomness_generation can pre-generate it)."
)

_default_engine = None


def default_engine():
    """The process-wide HIP engine (created on first use; raises when the library or the GPU is missing)."""
    global _default_engine
    if _default_engine is None:
        from .engine import Engine

        _default_engine = Engine()
    return _default_engine


def set_default_engine(engine) -> None:
    global _default_engine
    _default_engine = engine


def _fixed_point(x: Any, precision: int) -> int:
    """round(x * 10^precision) as an exact integer: ints scale exactly, floats go through their shortest decimal
    representation (so 0.1 with precision 1 is 1, not 0.1000000000000000055... rounded) with ties away from zero."""
    if isinstance(x, bool):
        x = int(x)
    if isinstance(x, int):
        return x * 10 ** precision
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            raise ValueError("cannot encode a non-finite plaintext")
        from decimal import ROUND_HALF_UP, Decimal

        scaled = Decimal(repr(x)).scaleb(precision)
        q = scaled.to_integral_value(rounding=ROUND_HALF_UP)
        if q != scaled:
            warnings.warn(f"plaintext {x!r} has more than {precision} decimal digits: rounded to {q}e-{precision}", UserWarning)
        return int(q)
    return int(x) * 10 ** precision


class _PublicKey:
    def __init__(self, **kw: int) -> None:
        self.__dict__.update(kw)

    def __eq__(self, other: object) -> bool:
        return isinstance(other, _PublicKey) and self.__dict__ == other.__dict__

    def __repr__(self) -> str:
        return f"PublicKey({', '.join(k for k in self.__dict__)})"


# =====================================================================================================
# ciphertext objects (single-ciphertext API of the reference)
# =====================================================================================================
class _Ciphertext:
    """A ciphertext: an integer modulo the scheme's ciphertext modulus, bound to its scheme.  The integer may be held as a ROW OF
    WORDS instead (`raw_value` = (array [rows][nwords] of uint32, row index)): what the batch launches of coalesced sessions hand
    out -- the Python integer is only made when somebody asks for it (`value`, `peek_value`, `get_value`)."""

    __slots__ = ("_raw_value", "_block", "_row", "scheme", "_fresh")      # (no per-object dict: a message holds l + 1 of these, a
                                                                         # thousand concurrent sessions a hundred thousand)

    def __init__(self, raw_value: Any, scheme: Any, *, fresh: bool = False) -> None:
        if isinstance(raw_value, tuple):
            self._raw_value, (self._block, self._row) = None, raw_value
        else:
            self._raw_value, self._block, self._row = int(raw_value), None, 0
        self.scheme = scheme
        self._fresh = fresh

    @classmethod
    def rows(cls, block, scheme: Any, fresh: bool = False, start: int = 0, count: int | None = None) -> list:
        """One ciphertext per row of `block` (an array [rows][nwords] of words, or a limbs.RowBlock), in order -- what a batch launch
        hands to a session; `start`, `count`: only that run of rows."""
        out = []
        for j in range(start, len(block) if count is None else start + count):
            c = object.__new__(cls)
            c._raw_value, c._block, c._row, c.scheme, c._fresh = None, block, j, scheme, fresh
            out.append(c)
        return out

    def _int(self) -> int:
        v = self._raw_value
        if v is None:
            blk = self._block
            row = blk.row(self._row) if type(blk) is RowBlock else blk[self._row]
            v = self._raw_value = int.from_bytes(row.tobytes(), "little")
        return v

    @property
    def value(self) -> int:
        """The ciphertext integer; reading it this way ends the ciphertext's freshness (it may have been observed)."""
        self._fresh = False
        return self._int()

    def peek_value(self) -> int:
        """The ciphertext integer without touching the freshness flag."""
        return self._int()

    def peek_words(self):
        """(array, row) when the value is held as a row of little-endian 32-bit words, else None."""
        return None if self._block is None else (self._block, self._row)

    def get_value(self) -> int:
        """Value for use in a homomorphic operation: warns when a fresh ciphertext is consumed that way."""
        if self._fresh:
            warnings.warn(WARN_INEFFICIENT_HOM_OPERATION, UserWarning)
        self._fresh = False
        return self._int()

    def consume(self) -> "_Ciphertext":
        """get_value's bookkeeping without making the integer (coalesced sessions pass word rows on)."""
        if self._fresh:
            warnings.warn(WARN_INEFFICIENT_HOM_OPERATION, UserWarning)
        self._fresh = False
        return self

    def for_wire(self):
        """What a transport should put on the wire: a fresh ciphertext (randomizing first, with a warning, if this one is
        not fresh); the sender's copy stops being fresh because it has been disclosed."""
        if not self._fresh:
            warnings.warn(WARN_UNFRESH_SERIALIZATION, UserWarning)
            self.randomize()
        self._fresh = False
        out = object.__new__(type(self))             # what arrives is bound to the PUBLIC scheme
        out._raw_value, out._block, out._row, out.scheme, out._fresh = self._raw_value, self._block, self._row, self.scheme.for_wire(), False
        return out

    @classmethod
    def wire_list(cls, cts: list) -> list:
        """for_wire() for a list of ciphertexts (a message of l + 1 of them): the same flags and bindings as calling it on each.  A fresh
        ciphertext that is bound to a public scheme already is handed over as it is (an in-memory transport moves objects; the sender
        of a message does not keep using them), everything else is copied onto the public scheme."""
        out, scheme, pub = [], None, None
        for c in cts:
            if type(c) is not cls or not c._fresh:
                out.append(c.for_wire())
                continue
            if c.scheme is not scheme:
                scheme, pub = c.scheme, c.scheme.for_wire()
            c._fresh = False
            if pub is scheme:
                out.append(c)
                continue
            w = object.__new__(cls)
            w._raw_value, w._block, w._row, w.scheme, w._fresh = c._raw_value, c._block, c._row, pub, False
            out.append(w)
        return out

    @property
    def fresh(self) -> bool:
        return self._fresh

    def randomize(self):
        """In-place re-randomization (SC/initiator.py:109,153-154; SC/keyholder.py:106-108,126-128)."""
        if self._fresh:
            warnings.warn(WARN_INEFFICIENT_RANDOMIZATION, UserWarning)
        self._raw_value, self._block = self.scheme._apply_randomness(self._int(), self.scheme.get_randomness()), None
        self._fresh = True
        return self

    def copy(self):
        return type(self)(self._int(), self.scheme)

    # ---- operator algebra
    def _coerce(self, other: Any):
        if isinstance(other, _Ciphertext):
            if other.scheme != self.scheme:
                raise ValueError("ciphertexts belong to different schemes")
            return other.get_value()
        return self.scheme._unsafe_encrypt_raw_value(self.scheme._encode(other))

    def __add__(self, other: Any):
        return type(self)(self.scheme._mul_values(self.get_value(), self._coerce(other)), self.scheme)

    __radd__ = __add__

    def __iadd__(self, other: Any):
        return self.__add__(other)

    def __neg__(self):
        return type(self)(self.scheme._inv_value(self.get_value()), self.scheme)

    def __sub__(self, other: Any):
        if isinstance(other, _Ciphertext):
            return self + (-other)
        return self + (-other)

    def __rsub__(self, other: Any):
        return (-self) + other

    def __mul__(self, scalar: int):
        if not isinstance(scalar, int):
            raise TypeError("ciphertexts can only be multiplied by integers")
        return type(self)(self.scheme._pow_value(self.get_value(), scalar), self.scheme)

    __rmul__ = __mul__

    def __imul__(self, scalar: int):
        return self.__mul__(scalar)

    def __eq__(self, other: object) -> bool:
        return isinstance(other, _Ciphertext) and self._int() == other._int() and self.scheme == other.scheme

    def __hash__(self) -> int:
        return hash((self._int(), id(self.scheme)))

    def __repr__(self) -> str:
        return f"<{type(self).__name__} {self._int():#x}>"


class PaillierCiphertext(_Ciphertext):
    """Paillier ciphertext: PaillierCiphertext(value, scheme) as in the reference's tests (:176)."""

    __slots__ = ()


class DGKCiphertext(_Ciphertext):
    """DGK ciphertext."""

    __slots__ = ()


# =====================================================================================================
# shared scheme machinery
# =====================================================================================================
class _Scheme:
    _ct_class = _Ciphertext

    def __init__(self, engine=None) -> None:
        self._engine = engine
        self._pool: list[int] = []
        self._pending: list = []                     # background generation jobs (boot_randomness_generation(background=True))
        self._background = None                      # (twin of this scheme on a second engine, its stream)
        self._wire_copy = None                       # the public copy a transport hands to the other party (for_wire)
        self._batch_pool: torch.Tensor | None = None

    @property
    def engine(self):
        if self._engine is None:
            self._engine = default_engine()
        return self._engine

    def for_wire(self):
        """What a transport puts on the wire for a scheme: its PUBLIC part (the reference's serializers drop the secret key).  One
        public copy per scheme object, so that everything a party receives from this scheme's owner is bound to the same object."""
        if getattr(self, "secret_key", None) is None:
            return self
        if self._wire_copy is None:
            self._wire_copy = self.public_copy()
        return self._wire_copy

    # ---- single-value helpers: one-element batches on the GPU
    def _one(self, value: int, nwords: int) -> torch.Tensor:
        return self.engine.upload([value], nwords)

    def _mul_values(self, a: int, b: int) -> int:
        m = self._ct_mod
        return self.engine.download(self.engine.modmul(m, self._one(a, m.nwords), self._one(b, m.nwords)))[0]

    def _inv_value(self, a: int) -> int:
        m = self._ct_mod
        return self.engine.download(self.engine.modinv(m, self._one(a, m.nwords)))[0]

    def _pow_value(self, a: int, k: int) -> int:
        """a^k mod n for a per-ciphertext scalar k (ct * k): the exponent travels as data (sc_modexp_var), so a long-running
        one-comparison-at-a-time service registers nothing per scalar -- only key-derived exponents are registered."""
        m = self._ct_mod
        if k < 0:
            a, k = self._inv_value(a), -k
        if k == 0:
            return 1 % m.n
        ew = (k.bit_length() + 31) // 32
        e = self.engine.upload([k], ew)
        return self.engine.download(self.engine.modexp_var(m, self._one(a, m.nwords), e, 32 * ew))[0]

    # ---- randomness pool (boot_randomness_generation / get_randomness / shut_down of the templates package)
    def boot_randomness_generation(self, amount: int, background: bool = False) -> None:
        """Pre-generate `amount` randomizers on the GPU (SC/initiator.py:209-210, SC/keyholder.py:178-179).
        background: like the reference's generation in background workers -- the exponentiations are queued on a second library
        context and stream of this scheme and the call returns; get_randomness collects them (waits, downloads) when the pool runs
        dry.  For randomizers that are not needed at once: the key holder's three Paillier randomizers of step 5 are computed
        while the initiator's first message is still being made.  Same pool, same order of use."""
        if amount <= 0:
            return
        if background:
            job = self._launch_randomness(amount)
            if job is not None:
                self._pending.append(job)
                return
        self._pool.extend(self._generate_randomness(amount))

    def _launch_randomness(self, amount: int):
        """(device array of `amount` finished randomizers, event) queued on the background context, or None where there is no
        second context to be had (schemes without a background twin, the CPU test tier's stand-in engine)."""
        return None

    def _launch_randomness_values(self, values: list[int]):
        """The same for randomizer inputs the caller drew itself (coalesced sessions draw their own)."""
        return None

    def _collect_pending(self) -> None:
        t, ev, twin = self._pending.pop(0)
        ev.synchronize()
        self._pool.extend(twin.engine.download(t))

    def get_randomness(self) -> int:
        if not self._pool and self._pending:
            self._collect_pending()
        if not self._pool:
            warnings.warn(WARN_OUT_OF_RANDOMNESS, UserWarning)
            self._pool.extend(self._generate_randomness(1))
        return self._pool.pop()

    def shut_down(self) -> None:
        """Stop the randomness generation (README.md:143-147 of the reference): what is still in flight is collected, the pools are
        dropped, and the background twin -- a second library context holding this scheme's key material, its tables and a stream of
        its own -- is released (a service that makes schemes per session would otherwise pile up contexts and hardware queues)."""
        while self._pending:
            self._collect_pending()
        self._pool.clear()
        self._batch_pool = None
        bg, self._background = self._background, None
        if bg is not None:
            twin, stream = bg
            stream.synchronize()
            eng = twin._engine
            twin._key = None
            if eng is not None and hasattr(eng, "close"):
                eng.close()

    # ---- device-resident randomizer pools for whole batches
    def boot_randomness_generation_batch(self, amount: int, source: str = "device", generator=None) -> None:
        """Pre-generate `amount` randomizers on the GPU and keep them as a device array (the batched analogue of
        boot_randomness_generation: the expensive exponentiations happen ahead of the protocol run)."""
        if amount <= 0:
            return
        fresh = self._generate_randomness_batch(amount, source, generator)
        pool = getattr(self, "_batch_pool", None)
        self._batch_pool = fresh if pool is None or pool.shape[0] == 0 else torch.cat([pool, fresh], dim=0)

    def take_randomness_batch(self, amount: int, source: str = "device", generator=None) -> torch.Tensor:
        pool = getattr(self, "_batch_pool", None)
        have = 0 if pool is None else pool.shape[0]
        if have < amount:
            warnings.warn(WARN_OUT_OF_RANDOMNESS, UserWarning)
            self.boot_randomness_generation_batch(amount - have, source, generator)
            pool = self._batch_pool
        out, self._batch_pool = pool[:amount], pool[amount:]
        return out

    def randomize_from_pool_batch(self, c: torch.Tensor) -> torch.Tensor:
        """`.randomize()` for a batch: one modular product with pooled randomizers."""
        return self.engine.modmul(self._ct_mod, c, self.take_randomness_batch(c.shape[0]))

    def __ne__(self, other: object) -> bool:
        return not self.__eq__(other)


# =====================================================================================================
# Paillier
# =====================================================================================================
class Paillier(_Scheme):
    """Paillier with g = N + 1.  `p`, `q` absent = public part only (Alice's copy)."""

    _ct_class = PaillierCiphertext

    def __init__(self, n: int, p: int | None = None, q: int | None = None, engine=None, use_crt: bool = True,
                 use_pairs: bool = True, precision: int = 0) -> None:
        super().__init__(engine)
        self.use_pairs = use_pairs  # exponentiations modulo N^2 / p^2 / q^2 through pair arithmetic modulo N / p / q
        # decimal digits kept of a non-integral plaintext: x is encoded as round(x * 10^precision) ([ext] fixed-point
        # encoding of the scheme package; perform_secure_comparison advertises `PaillierCiphertext | float`, SC/initiator.py:69-72)
        self.precision = int(precision)
        self.public_key = _PublicKey(n=n, n_squared=n * n, g=n + 1)
        self.secret_key = None
        self.use_crt = use_crt
        if p is not None and q is not None:
            if p * q != n:
                raise ValueError("p * q != n")
            lam = (p - 1) * (q - 1)
            self.secret_key = _PublicKey(p=p, q=q, lambda_=lam, mu=pow(lam, -1, n))
        self._key = None

    @classmethod
    def from_security_parameter(cls, key_length: int = 2048, engine=None, precision: int = 0, **_ignored: Any) -> "Paillier":
        """Fresh key pair (SC/keyholder.py:156); prime generation runs on the host."""
        p, q = keygen.paillier_primes(key_length)
        return cls(p * q, p, q, engine=engine, precision=precision)

    def public_copy(self) -> "Paillier":
        return Paillier(self.public_key.n, engine=self._engine, precision=self.precision)

    def __eq__(self, other: object) -> bool:
        return isinstance(other, Paillier) and self.public_key.n == other.public_key.n

    __hash__ = None  # type: ignore[assignment]

    # ---- engine handles
    @property
    def key(self):
        """The library-side key object (sc_paillier_key_create): moduli, exponents, CRT constants -- derived once, in the library."""
        if self._key is None:
            sk = self.secret_key
            self._key = self.engine.paillier_key(self.public_key.n, None if sk is None else sk.p, None if sk is None else sk.q,
                                                 use_crt=self.use_crt, use_pairs=self.use_pairs)
        return self._key

    @property
    def mod_n(self):
        return self.key.mod_n

    @property
    def mod_n2(self):
        return self.key.mod_n2

    @property
    def _ct_mod(self):
        return self.mod_n2

    # ---- encoding: signed fixed point with `precision` decimal digits; negatives wrap mod N ([ext] default encoding)
    def _encode(self, m: Any) -> int:
        return _fixed_point(m, self.precision) % self.public_key.n

    def _decode(self, m: int):
        v = m - self.public_key.n if m > self.public_key.n // 2 else m
        return v if self.precision == 0 else v / 10 ** self.precision

    # ---- batched API (device tensors [count][words]): one library call each
    def encrypt_raw_batch(self, m_words: torch.Tensor) -> torch.Tensor:
        """[[m]] = 1 + m N mod N^2 without randomness, for plaintext words [count][<= 2*nw]."""
        return self.engine.paillier_encrypt(self.key, m_words)

    def encrypt_raw_neg_batch(self, m_words: torch.Tensor) -> torch.Tensor:
        """[[-m]] = 1 - m N mod N^2 = ([[m]])^-1, with no modular inversion."""
        return self.engine.paillier_encrypt(self.key, m_words, negate=True)

    def randomizer_batch(self, rho: torch.Tensor) -> torch.Tensor:
        """rho^N mod N^2 for rho words [count][nw(N)]."""
        return self.randomize_batch(None, rho)

    def randomize_batch(self, c: torch.Tensor | None, rho: torch.Tensor) -> torch.Tensor:
        """c * rho^N mod N^2 (c = None: just the randomizer): sc_paillier_randomize.  The key holder's key goes through CRT over
        p^2, q^2 inside the library (identical integers); Alice's through the pair arithmetic modulo N."""
        return self.engine.paillier_randomize(self.key, c, rho)

    def decrypt_raw_batch(self, c: torch.Tensor) -> torch.Tensor:
        """m = L(c^lambda mod N^2) mu mod N, words [count][nw(N)] (SC/keyholder.py:195): sc_paillier_decrypt."""
        if self.secret_key is None:
            raise ValueError("this Paillier scheme has no secret key")
        return self.engine.paillier_decrypt(self.key, c)

    def add_batch(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        return self.engine.modmul(self.mod_n2, a, b)

    def neg_batch(self, a: torch.Tensor) -> torch.Tensor:
        return self.engine.modinv(self.mod_n2, a)

    # ---- single-ciphertext API of the reference
    def _unsafe_encrypt_raw_value(self, m: int) -> int:
        nw = self.mod_n.nwords
        mm = m % self.public_key.n
        return self.engine.download(self.encrypt_raw_batch(self._one(mm, nw)))[0]

    def unsafe_encrypt(self, plaintext: Any, apply_encoding: bool = True) -> PaillierCiphertext:
        m = self._encode(plaintext) if apply_encoding else int(plaintext)
        return PaillierCiphertext(self._unsafe_encrypt_raw_value(m), self)

    def encrypt(self, plaintext: Any, apply_encoding: bool = True) -> PaillierCiphertext:
        ct = self.unsafe_encrypt(plaintext, apply_encoding)
        ct.randomize()
        return ct

    def decrypt(self, ciphertext: PaillierCiphertext, apply_encoding: bool = True) -> int:
        m = self.engine.download(self.decrypt_raw_batch(self._one(ciphertext.peek_value(), self.mod_n2.nwords)))[0]
        return self._decode(m) if apply_encoding else m

    def _generate_randomness(self, amount: int) -> list[int]:
        n = self.public_key.n
        rho = self.engine.upload([1 + secrets.randbelow(n - 1) for _ in range(amount)], self.mod_n.nwords)
        return self.engine.download(self.randomizer_batch(rho))

    def _generate_randomness_batch(self, amount: int, source: str = "device", generator=None) -> torch.Tensor:
        from .randomness import uniform_below

        rho = uniform_below(self.public_key.n, amount, self.engine, source, generator, nonzero=True)
        return self.randomizer_batch(rho)

    def _launch_randomness(self, amount: int):
        from .engine import Engine

        if not isinstance(self.engine, Engine):      # the CPU test tier's stand-in engine: nothing to overlap (and nothing drawn here)
            return None
        n = self.public_key.n
        return self._launch_randomness_values([1 + secrets.randbelow(n - 1) for _ in range(amount)])      # drawn now, in the caller's order

    def _launch_randomness_values(self, values: list[int]):
        from .engine import Engine

        eng = self.engine
        if not isinstance(eng, Engine):
            return None
        if self._background is None:
            sk = self.secret_key
            twin = Paillier(self.public_key.n, sk.p if sk else None, sk.q if sk else None, engine=Engine(eng.device_index),
                            use_crt=self.use_crt, use_pairs=self.use_pairs, precision=self.precision)
            self._background = (twin, torch.cuda.Stream(device=eng.device))
        twin, stream = self._background
        stream.wait_stream(torch.cuda.current_stream(eng.device))
        with torch.cuda.stream(stream):
            t = twin.randomizer_batch(twin.engine.upload(values, twin.mod_n.nwords))
            ev = torch.cuda.Event()
            ev.record(stream)
        return t, ev, twin

    def _apply_randomness(self, value: int, randomness: int) -> int:
        return self._mul_values(value, randomness)


# =====================================================================================================
# DGK
# =====================================================================================================
class DGK(_Scheme):
    """DGK scheme: Enc(m) = g^m h^r mod n; zero test c^{v_p} mod p == 1 (SC/keyholder.py:249)."""

    _ct_class = DGKCiphertext

    def __init__(self, n: int, g: int, h: int, u: int, t: int, p: int | None = None, q: int | None = None,
                 v_p: int | None = None, v_q: int | None = None, full_decryption: bool = False, engine=None,
                 randomizer_bits: int | None = None, fixed_base_window: int = 8, use_crt: bool = True) -> None:
        super().__init__(engine)
        self.use_crt = use_crt  # key holder only: randomizers h^r through CRT with exponents reduced modulo v_p, v_q
        self.public_key = _PublicKey(n=n, g=g, h=h, u=u, t=t)
        self.secret_key = None
        if p is not None:
            self.secret_key = _PublicKey(p=p, q=q, v_p=v_p, v_q=v_q)
        self.full_decryption = full_decryption
        self.randomizer_bits = randomizer_bits if randomizer_bits is not None else int(2.5 * t)
        self.fixed_base_window = fixed_base_window
        self._key = None
        self._table_source: "DGK | None" = None
        self.table_build_s = 0.0
        self._g_inv = None
        self._dec_table: dict[int, int] | None = None

    @classmethod
    def from_security_parameter(cls, v_bits: int = 160, n_bits: int = 2048, u: int = 2 ** 16 + 1,
                                full_decryption: bool = False, engine=None, **_ignored: Any) -> "DGK":
        """Fresh DGK key (SC/keyholder.py:161-166); key generation runs on the host."""
        k = keygen.dgk_key(v_bits, n_bits, u)
        return cls(k["n"], k["g"], k["h"], k["u"], k["t"], k["p"], k["q"], k["v_p"], k["v_q"], full_decryption, engine)

    def public_copy(self) -> "DGK":
        pk = self.public_key
        return DGK(pk.n, pk.g, pk.h, pk.u, pk.t, engine=self._engine, randomizer_bits=self.randomizer_bits,
                   fixed_base_window=self.fixed_base_window)

    def __eq__(self, other: object) -> bool:
        return isinstance(other, DGK) and self.public_key == other.public_key

    __hash__ = None  # type: ignore[assignment]

    @property
    def key(self):
        """The library-side key object (sc_dgk_key_create): moduli, g^-1, the CRT halves and the fixed-base tables for h --
        built on first use, or taken over read-only from the scheme object named by share_tables_from."""
        if self._key is None:
            import time

            pk, sk, src = self.public_key, self.secret_key, self._table_source
            t0 = time.perf_counter()
            self._key = self.engine.dgk_key(pk.n, pk.g, pk.h, pk.u, pk.t, *((None,) * 4 if sk is None else (sk.p, sk.q, sk.v_p, sk.v_q)),
                                            randomizer_bits=self.randomizer_bits, window=self.fixed_base_window, use_crt=self.use_crt,
                                            table_source=None if src is None else (src.engine, src.key))
            self.engine.synchronize()
            if src is None:
                self.table_build_s += time.perf_counter() - t0
        return self._key

    def prepare(self) -> "DGK":
        """Build (or import) the key object and its tables now -- untimed set-up, like key generation."""
        _ = self.key
        return self

    @property
    def mod_n(self):
        return self.key.mod_n

    @property
    def mod_p(self):
        if self.secret_key is None:
            raise ValueError("this DGK scheme has no secret key")
        return self.key.mod_p

    @property
    def _ct_mod(self):
        return self.mod_n

    @property
    def g_inv(self) -> int:
        if self._g_inv is None:
            self._g_inv = pow(self.public_key.g, -1, self.public_key.n)     # set-up constant (the library derives its own copy)
        return self._g_inv

    def share_tables_from(self, other: "DGK") -> None:
        """Read `other`'s device-resident fixed-base tables instead of building copies (same key, window and randomizer
        width; `other` may be bound to another engine of the same GPU -- the concurrent shards of batch.ConcurrentShards)."""
        if other.public_key != self.public_key or other.fixed_base_window != self.fixed_base_window or \
                other.randomizer_bits != self.randomizer_bits or (other.secret_key is None) != (self.secret_key is None) or \
                other.use_crt != self.use_crt:
            raise ValueError("tables can only be shared between scheme objects of the same key and table parameters")
        self._table_source = other

    def table_bytes(self) -> int:
        """Device bytes of the fixed-base tables of this object's key (built or taken over)."""
        return 0 if self._key is None else self.engine.dgk_table_bytes(self._key)

    def _encode(self, m: Any) -> int:
        if isinstance(m, float):
            if not m.is_integer():
                raise ValueError("DGK plaintexts are integers modulo u")
            m = int(m)
        return int(m)

    # ---- batched API: one library call each
    def encrypt_bits_batch(self, bits: torch.Tensor) -> torch.Tensor:
        """g^b for b in {0,1}: words [count][nw] (SC/keyholder.py:213, 231)."""
        e = self.engine
        if getattr(self, "_bit_words", None) is None:      # [0] = g^0 = 1, [1] = g: uploaded once per scheme object
            self._bit_words = e.upload([1, self.public_key.g], self.mod_n.nwords)
        return self._bit_words[(bits != 0).reshape(-1).to(torch.int64)]

    def encrypt_bits_randomized_batch(self, bits: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
        """g^b * h^r for bits b [count] and exponent words r [count][ewords]: `unsafe_encrypt(bit)` and `.randomize()` of
        SC/keyholder.py:213, 231 and :106-108 in one library call (sc_dgk_encrypt_bits_randomized)."""
        return self.engine.dgk_encrypt_bits_randomized(self.key, (bits.reshape(-1) != 0).to(torch.uint8), r)

    def randomize_batch(self, c: torch.Tensor | None, r: torch.Tensor) -> torch.Tensor:
        """c * h^r mod n for exponent words r [count][ewords] (SC/keyholder.py:106-108; SC/initiator.py:153-154):
        sc_dgk_randomize.  The key holder's key goes through CRT inside the library (h has order v_p modulo p: a 160-bit
        exponent and a half-size modulus per prime); identical residues."""
        return self.engine.dgk_randomize(self.key, c, r)

    def is_zero_batch(self, c: torch.Tensor) -> torch.Tensor:
        """uint8 flags: plaintext == 0 mod u (SC/keyholder.py:249)."""
        return self.engine.dgk_is_zero(self.key, c)

    def any_zero_batch(self, c: torch.Tensor) -> torch.Tensor:
        """int64 [B]: 1 where some plane of the bit-major vector c [planes][B][nw] decrypts to 0 mod u (KeyHolder.step_4j)."""
        return self.engine.dgk_any_zero(self.key, c)

    def neg_batch(self, c: torch.Tensor) -> torch.Tensor:
        return self.engine.modinv(self.mod_n, c)

    # ---- single-ciphertext API
    def _unsafe_encrypt_raw_value(self, m: int) -> int:
        g = self.public_key.g
        if m == 0:
            return 1
        if m == 1:
            return g % self.public_key.n
        return self._pow_value(g, m)

    def unsafe_encrypt(self, plaintext: Any, apply_encoding: bool = True) -> DGKCiphertext:
        m = self._encode(plaintext) if apply_encoding else int(plaintext)
        return DGKCiphertext(self._unsafe_encrypt_raw_value(m), self)

    def encrypt(self, plaintext: Any, apply_encoding: bool = True) -> DGKCiphertext:
        ct = self.unsafe_encrypt(plaintext, apply_encoding)
        ct.randomize()
        return ct

    def is_zero(self, ciphertext: DGKCiphertext) -> bool:
        flags = self.is_zero_batch(self._one(ciphertext.peek_value(), self.mod_n.nwords))
        return bool(flags.cpu()[0])

    def decrypt(self, ciphertext: DGKCiphertext, apply_encoding: bool = True) -> int:
        """Full decryption through a host lookup table (tests only; needs full_decryption=True and a small u)."""
        if not self.full_decryption or self.secret_key is None:
            raise ValueError("full decryption is not enabled for this DGK scheme")
        sk, pk = self.secret_key, self.public_key
        if self._dec_table is None:
            gv = pow(pk.g % sk.p, sk.v_p, sk.p)
            table, acc = {}, 1
            for m in range(pk.u):
                table[acc] = m
                acc = acc * gv % sk.p
            self._dec_table = table
        x = self.engine.download(self.engine.modexp_shared(self.mod_p, self._one(ciphertext.peek_value(), self.mod_n.nwords), sk.v_p))[0]
        m = self._dec_table[x]
        if apply_encoding and m > pk.u // 2:
            m -= pk.u
        return m

    def _generate_randomness(self, amount: int) -> list[int]:
        ew = (self.randomizer_bits + 31) // 32
        r = self.engine.upload([secrets.randbits(self.randomizer_bits) for _ in range(amount)], ew)
        return self.engine.download(self.randomize_batch(None, r))

    def _generate_randomness_batch(self, amount: int, source: str = "device", generator=None) -> torch.Tensor:
        from .randomness import random_bits

        return self.randomize_batch(None, random_bits(self.randomizer_bits, (amount,), self.engine, source, generator))

    def _apply_randomness(self, value: int, randomness: int) -> int:
        return self._mul_values(value, randomness)
