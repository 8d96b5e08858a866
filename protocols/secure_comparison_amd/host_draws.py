"""The random draws of ONE interactive comparison on the host.

The reference draws every value with `secrets` (SC/initiator.py:223 permutation, :250 r, :420 delta_A, :512 rho_i; the scheme
packages' randomizer inputs behind boot_randomness_generation, :205-210, SC/keyholder.py:174-179): one system call and one Python
integer per draw -- about a hundred draws and 0.2 ms per comparison, nothing beside 100 ms of CPU arithmetic, but more than the GPU
spends on the comparison when a thousand concurrent sessions share its launches (coalesce.py).  A session therefore takes its draws
through a `HostDraws` object, in bulk where a step consumes them in bulk, and as little-endian 32-bit word rows -- the form the
library takes them in -- where the integer itself is never needed:

* `HostDraws` (default): a buffered stream from the operating system's generator (os.urandom, what `secrets` reads) or from the
  engine's device-side CSPRNG, rejection sampling vectorised with numpy;
* `SecretsDraws`: every value through `secrets.randbelow` / `secrets.randbits`, one call per value, in the order the reference makes
  them -- what the single (uncoalesced) path does, and what the tests inject to replay a session under a seeded stream.

Both return the same formats, so the sessions' arithmetic never knows which one drew.
"""
from __future__ import annotations

import os
import secrets
import weakref

import numpy as np


_instances: "weakref.WeakSet[HostDraws]" = weakref.WeakSet()


def _forget_all_in_child() -> None:
    # a forked child must never hand out the values its parent still holds in its buffers: a randomizer used twice gives the
    # difference of two plaintexts away (`secrets` reads the OS generator per call and has no such state)
    for d in list(_instances):
        d.forget()


os.register_at_fork(after_in_child=_forget_all_in_child)


class HostDraws:
    """Draws from a buffered stream of random bytes.  `refill(nbytes) -> bytes` supplies the stream: os.urandom by default (what
    `secrets` reads), or the engine's device generator (ChaCha20 keyed from the OS, csrc/sc_rng.h) through `from_engine` -- the
    host's generator delivers ~0.35 GB/s, and a thousand concurrent sessions consume 5 MB.  Word rows are uint32, least
    significant word first."""

    CHUNK = 1 << 20

    def __init__(self, refill=None) -> None:
        self._refill = refill if refill is not None else os.urandom
        self._buf, self._at = b"", 0
        _instances.add(self)

    def forget(self) -> None:
        """Drop everything drawn ahead (the buffered stream and the pools)."""
        self._buf, self._at = b"", 0
        for name in ("_bits_pools", "_below_pools", "_perm_pools"):
            self.__dict__.pop(name, None)

    @classmethod
    def from_engine(cls, engine) -> "HostDraws":
        """The stream of the engine's device-side CSPRNG where the engine has one (the GPU engine), else the OS generator."""
        from .engine import Engine

        if not isinstance(engine, Engine):          # (the CPU test tier's stand-in engine restates the generator in pure Python)
            return cls()

        def refill(nbytes: int) -> bytes:
            return engine.download_words(engine.rng_bits(32, (nbytes + 3) // 4)).tobytes()

        return cls(refill)

    def take(self, nbytes: int) -> bytes:
        at = self._at
        if at + nbytes > len(self._buf):
            self._buf, at = self._refill(max(self.CHUNK, nbytes)), 0
        self._at = at + nbytes
        return self._buf[at:at + nbytes]

    def randbelow(self, n: int) -> int:
        """Uniform in [0, n)."""
        k = n.bit_length()
        nbytes, mask = (k + 7) // 8, (1 << k) - 1
        while True:
            v = int.from_bytes(self.take(nbytes), "little") & mask
            if v < n:
                return v

    def bits_rows(self, bits: int, count: int) -> np.ndarray:
        """`count` uniform integers below 2^bits as rows [count][ceil(bits/32)] (the DGK randomizer exponents); pooled like
        below_rows_nonzero (the rows handed out are views of the pool: read them, do not write them)."""
        pools = self.__dict__.setdefault("_bits_pools", {})
        rows, at = pools.get(bits, (None, 0))
        if rows is None or at + count > len(rows):
            n = max(self.POOL_ROWS, count)
            nw = (bits + 31) // 32
            rows, at = np.frombuffer(self.take(4 * nw * n), dtype="<u4").reshape(n, nw).copy(), 0
            top = bits - 32 * (nw - 1)
            if top < 32:
                rows[:, -1] &= np.uint32((1 << top) - 1)
        pools[bits] = (rows, at + count)
        return rows[at:at + count]

    POOL_ROWS = 8192          # values / permutations drawn per vectorised refill of a pool

    def below_rows_nonzero(self, n: int, count: int) -> np.ndarray:
        """`count` uniform integers in [1, n) as rows [count][ceil(bitlen(n)/32)], n < 2^127 (step 4i's rho_i = 1 + randbelow(u - 1)).
        Independent draws are handed out from a pool that is refilled a few thousand values at a time: the numpy calls of a
        refill cost the same for 33 values as for 8192, and a session asks for l + 1."""
        pools = self.__dict__.setdefault("_below_pools", {})
        rows, at = pools.get(n, (None, 0))
        if rows is None or at + count > len(rows):
            rows, at = self._below_rows_nonzero(n, max(self.POOL_ROWS, count)), 0
        pools[n] = (rows, at + count)
        return rows[at:at + count]

    def _below_rows_nonzero(self, n: int, count: int) -> np.ndarray:
        m = n - 1                                  # draw v in [0, m), return v + 1
        k = m.bit_length()
        nw = (n.bit_length() + 31) // 32
        if k <= 62:                                # one 64-bit column, no carry out of v + 1 (every l <= 60)
            got = np.empty(0, dtype="<u8")
            while len(got) < count:
                cand = np.frombuffer(self.take(8 * (2 * (count - len(got)) + 8)), dtype="<u8") & np.uint64((1 << k) - 1)
                got = np.concatenate([got, cand[cand < np.uint64(m)]])
            return (got[:count] + np.uint64(1)).view("<u4").reshape(count, 2)[:, :nw].copy()
        if k > 126:
            raise ValueError("below_rows_nonzero: bounds of at most 126 bits")
        out = np.empty((count, nw), dtype="<u4")
        filled = 0
        lo_mask, hi_bits = (1 << 64) - 1, k - 64
        m_lo, m_hi = np.uint64(m & ((1 << 64) - 1)), np.uint64(m >> 64)
        while filled < count:
            need = max(8, 2 * (count - filled) + 4)
            raw = np.frombuffer(self.take(16 * need), dtype="<u8").reshape(need, 2)
            lo = raw[:, 0] & np.uint64(lo_mask)
            hi = raw[:, 1] & np.uint64((1 << hi_bits) - 1) if hi_bits > 0 else np.zeros(need, dtype="<u8")
            ok = (hi < m_hi) | ((hi == m_hi) & (lo < m_lo))
            lo, hi = lo[ok][: count - filled], hi[ok][: count - filled]
            lo1 = lo + np.uint64(1)                                  # + 1 with the carry into the high half
            hi1 = hi + (lo1 == 0).astype("<u8")
            words = np.stack([lo1 & np.uint64(0xFFFFFFFF), lo1 >> np.uint64(32), hi1 & np.uint64(0xFFFFFFFF), hi1 >> np.uint64(32)], axis=1).astype("<u4")
            out[filled:filled + len(lo)] = words[:, :nw]
            filled += len(lo)
        return out

    def coin(self) -> int:
        return self.take(1)[0] & 1

    def permutation(self, k: int) -> list[int]:
        """Uniform permutation of range(k): entry j = the source index of output j (the shuffle of SC/initiator.py:212-226).  Sorting
        k independent 64-bit keys orders them uniformly; a row with a tie (probability < k^2 / 2^65) is dropped, so every
        permutation handed out is exactly uniform.  Pooled like below_rows_nonzero."""
        pools = self.__dict__.setdefault("_perm_pools", {})
        perms = pools.get(k)
        if not perms:
            rows = max(64, self.POOL_ROWS // max(1, k))
            keys = np.frombuffer(self.take(8 * k * rows), dtype="<u8").reshape(rows, k)
            order = np.argsort(keys, axis=1, kind="stable")
            if k > 1:
                ok = (np.diff(np.take_along_axis(keys, order, axis=1), axis=1) != 0).all(axis=1)
                order = order[ok]
            perms = pools[k] = order.tolist()
        return perms.pop()


class SecretsDraws(HostDraws):
    """Every value through `secrets`, one call per value, in the reference's order."""

    def __init__(self) -> None:
        super().__init__(None)

    def randbelow(self, n: int) -> int:
        return secrets.randbelow(n)

    def bits_rows(self, bits: int, count: int) -> np.ndarray:
        nw = (bits + 31) // 32
        buf = b"".join(secrets.randbits(bits).to_bytes(4 * nw, "little") for _ in range(count))
        return np.frombuffer(buf, dtype="<u4").reshape(count, nw).copy()

    def below_rows_nonzero(self, n: int, count: int) -> np.ndarray:
        nw = (n.bit_length() + 31) // 32
        buf = b"".join((secrets.randbelow(n - 1) + 1).to_bytes(4 * nw, "little") for _ in range(count))
        return np.frombuffer(buf, dtype="<u4").reshape(count, nw).copy()

    def coin(self) -> int:
        return secrets.randbelow(2)

    def permutation(self, k: int) -> list[int]:
        perm = list(range(k))
        for j in range(k - 1, 0, -1):
            i = secrets.randbelow(j + 1)
            perm[j], perm[i] = perm[i], perm[j]
        return perm
