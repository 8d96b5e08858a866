"""Host-side key generation (stays on the CPU; not part of the batched hot path).

Paillier: N = p q with g = N + 1.  DGK (SC/keyholder.py:161-166 parameters): primes v_p, v_q of v_bits,
p = 2 u v_p p_r + 1, q = 2 u v_q q_r + 1, g of order u v_p v_q, h of order v_p v_q.
"""
from __future__ import annotations

import secrets
from typing import Callable, Sequence

_SMALL = [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 101, 103, 107, 109, 113]


def is_prime(n: int, rounds: int = 32) -> bool:
    if n < 2:
        return False
    for p in _SMALL:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for i in range(rounds):
        a = _SMALL[i] if i < 8 else 2 + secrets.randbelow(n - 3)
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
    """Smallest prime larger than n (the reference imports this from tno.mpc.encryption_schemes.utils)."""
    c = max(n + 1, 2)
    if c > 2 and c % 2 == 0:
        c += 1
    while not is_prime(c):
        c += 1 if c == 2 else 2
    return c


def rand_prime(bits: int, randbits: Callable[[int], int] = secrets.randbits) -> int:
    while True:
        c = randbits(bits) | (1 << (bits - 1)) | 1
        if is_prime(c):
            return c


def paillier_primes(key_length: int) -> tuple[int, int]:
    while True:
        p = rand_prime(key_length // 2)
        q = rand_prime(key_length - key_length // 2)
        if p != q and (p * q).bit_length() == key_length:
            return p, q


def _elem_of_order(prime: int, cofactor: int, factors: Sequence[int]) -> int:
    order = 1
    for f in factors:
        order *= f
    while True:
        e = pow(2 + secrets.randbelow(prime - 3), cofactor, prime)
        if e != 1 and all(pow(e, order // f, prime) != 1 for f in factors):
            return e


def dgk_key(v_bits: int, n_bits: int, u: int) -> dict[str, int]:
    half = n_bits // 2
    v_p = rand_prime(v_bits)
    v_q = rand_prime(v_bits)
    while v_q == v_p:
        v_q = rand_prime(v_bits)

    def make(v: int, bits: int) -> tuple[int, int]:
        base = 2 * u * v
        need = bits - base.bit_length()
        if need <= 8:
            raise ValueError("n_bits too small for the requested u and v_bits")
        while True:
            cof = secrets.randbits(need + 1) | 1
            cand = base * cof + 1
            if cand.bit_length() == bits and is_prime(cand):
                return cand, cof

    while True:
        p, p_r = make(v_p, half)
        q, q_r = make(v_q, n_bits - half)
        if p != q and (p * q).bit_length() == n_bits:
            break
    n = p * q
    g_p, g_q = _elem_of_order(p, 2 * p_r, (u, v_p)), _elem_of_order(q, 2 * q_r, (u, v_q))
    h_p, h_q = _elem_of_order(p, 2 * p_r * u, (v_p,)), _elem_of_order(q, 2 * q_r * u, (v_q,))
    q_inv = pow(q, -1, p)

    def crt(a_p: int, a_q: int) -> int:
        return (a_q + q * ((a_p - a_q) * q_inv % p)) % n

    return {"n": n, "g": crt(g_p, g_q), "h": crt(h_p, h_q), "u": u, "t": v_bits, "p": p, "q": q, "v_p": v_p, "v_q": v_q}
