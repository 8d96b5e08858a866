"""TEST INFRASTRUCTURE -- integer-level model of the library's inversion kernel (csrc/sc_xgcd.h::k_xgcd).

Only tests/ may import this module.  It restates the kernel step for step in Python integers: the 30 division steps simulated on
the low 32-bit words (same wrap-around arithmetic as the scalar code), the lane-redundant representation (lane L holds a signed
value x_L, the number is sum x_L B^L, B = 2^(32 WPL)), the lane-local application of the 2x2 matrix with the 30 bits handed to
the lane below, and the final normalisation.  It exists to check the kernel's PROVEN bound on the lane values
(M_t < (t + 1)(2B + 1), see the kernel's header) on adversarial inputs, and that the model -- hence the algorithm -- returns
pow(x, -1, n).  The modular inverse serves the `- ct` / `ct * -1` operations of SC/initiator.py:254, :320, :371, :466, :478, :531, :559.
"""
from __future__ import annotations

M32 = 0xFFFFFFFF
LANES = 64


def _s32(v: int) -> int:
    v &= M32
    return v - (1 << 32) if v >> 31 else v


def _ctz(v: int) -> int:
    return (v & -v).bit_length() - 1


def divsteps30(eta: int, f: int, g: int) -> tuple[int, int, int, int, int]:
    """30 Bernstein-Yang division steps on the low words f, g (uint32); returns (eta, u, v, q, r) with signed 32-bit entries."""
    u, v, q, r = 1, 0, 0, 1
    i = 30
    while True:
        zeros = _ctz((g | (M32 << i)) & M32)
        g >>= zeros
        u, v = (u << zeros) & M32, (v << zeros) & M32
        eta -= zeros
        i -= zeros
        if i == 0:
            break
        if eta < 0:
            eta = -eta
            f, g = g, (-f) & M32
            u, q = q, (-u) & M32
            v, r = r, (-v) & M32
        limit = min(eta + 1, i)
        m = (M32 >> (32 - limit)) & 255
        x = f
        x = (x * ((2 - f * x) & M32)) & M32
        x = (x * ((2 - f * x) & M32)) & M32
        w = (g * ((-x) & M32)) & M32 & m
        g = (g + f * w) & M32
        q = (q + u * w) & M32
        r = (r + v * w) & M32
    return eta, _s32(u), _s32(v), _s32(q), _s32(r)


def to_lanes(x: int, wpl: int) -> list[int]:
    B = 1 << (32 * wpl)
    return [(x >> (32 * wpl * L)) & (B - 1) for L in range(LANES)]


def value(lanes: list[int], wpl: int) -> int:
    return sum(v << (32 * wpl * L) for L, v in enumerate(lanes))


def apply(a: int, X: list[int], b: int, Y: list[int], m: int, N: list[int], wpl: int) -> list[int]:
    """DS::apply for all lanes: y_L = a X_L + b Y_L + m N_L, x'_L = floor(y_L / 2^30) + (y_(L+1) mod 2^30) 2^(32 wpl - 30)."""
    y = [a * X[L] + b * Y[L] + m * N[L] for L in range(LANES)]
    shift = 32 * wpl - 30
    return [(y[L] >> 30) + (((y[L + 1] & 0x3FFFFFFF) << shift) if L + 1 < LANES else 0) for L in range(LANES)]


def rounds_for(nwords: int) -> int:
    bits = 32 * nwords
    return ((49 * bits + 57) // 17 + 1 + 29) // 30


def modinv(x: int, n: int, nwords: int, wpl: int, stats: dict | None = None) -> int | None:
    """The kernel's computation for one residue; None when x is not invertible.  stats (optional) receives the largest lane
    magnitude seen for (f, g) and (d, e) per round -- in units of B -- and the largest |u| + |v|, |q| + |r|."""
    assert nwords + 2 <= LANES * wpl and n & 1 and 0 <= x < n
    B = 1 << (32 * wpl)
    N = to_lanes(n, wpl)
    f, g, d, e = to_lanes(n, wpl), to_lanes(x, wpl), [0] * LANES, [1] + [0] * (LANES - 1)
    ninv = pow(n, -1, 1 << 30)
    eta = -1
    worst_fg = worst_de = worst_row = 0.0
    for t in range(rounds_for(nwords)):
        if not any(g):
            break
        eta, u, v, q, r = divsteps30(eta, f[0] & M32, g[0] & M32)
        cd, ce = (u * d[0] + v * e[0]) & 0x3FFFFFFF, (q * d[0] + r * e[0]) & 0x3FFFFFFF
        md, me = -((ninv * cd) & 0x3FFFFFFF), -((ninv * ce) & 0x3FFFFFFF)
        f, g, d, e = apply(u, f, v, g, 0, N, wpl), apply(q, f, r, g, 0, N, wpl), apply(u, d, v, e, md, N, wpl), apply(q, d, r, e, me, N, wpl)
        mfg, mde = max(abs(v_) for v_ in f + g), max(abs(v_) for v_ in d + e)
        assert mfg < (t + 2) * (B + 1), ("f/g lane bound", t)           # M_t < (t + 1)(B + 1) after t rounds
        assert mde < (t + 2) * (2 * B + 1), ("d/e lane bound", t)
        worst_fg, worst_de = max(worst_fg, mfg / B), max(worst_de, mde / B)
        worst_row = max(worst_row, abs(u) + abs(v), abs(q) + abs(r))
    if stats is not None:
        stats.update(worst_fg_in_B=worst_fg, worst_de_in_B=worst_de, worst_row_sum=worst_row)
    fv, gv, dv = value(f, wpl), value(g, wpl), value(d, wpl)
    if gv != 0 or fv not in (1, -1):
        return None
    return (dv if fv == 1 else -dv) % n
