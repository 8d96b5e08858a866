"""The inversion kernel's proven bound, checked on its integer-level model (oracle/xgcd_model.py restates csrc/sc_xgcd.h): the
lane values of the redundant representation grow ADDITIVELY -- below (t + 1)(B + 1) for (f, g) and (t + 1)(2B + 1) for (d, e) after
t rounds -- never a bit per round, for adversarial operands and for adversarial matrices fed to the lane step directly."""
import random

import pytest

from oracle import xgcd_model as xm


def _adversarial_pairs(bits, rng):
    n_all_ones = (1 << bits) - 1
    while n_all_ones % 3 == 0 or n_all_ones % 5 == 0:
        n_all_ones -= 2
    alt = int("a" * (bits // 4), 16) | 1                      # 1010...1011
    ns = [n_all_ones, (1 << (bits - 1)) | 1, alt | (1 << (bits - 1)), rng.getrandbits(bits) | (1 << (bits - 1)) | 1]
    for n in ns:
        for x in (1, 2, n - 1, n - 2, (n + 1) // 2, n >> 1, (1 << (bits - 2)) - 1, int("5" * (bits // 4 - 1), 16) % n, rng.randrange(1, n)):
            yield n, x % n


@pytest.mark.parametrize("bits, wpl", [(64, 1), (256, 1), (1024, 1), (1984, 1), (2048, 2), (4032, 2), (4096, 4)])
def test_model_inverts_and_stays_inside_the_proven_bound(bits, wpl):
    """modinv() asserts the bound after every round; here it runs on all-ones / single-bit / alternating moduli and on operands
    chosen to keep the division steps adding or exchanging for long stretches."""
    import math

    rng = random.Random(bits)
    nwords = bits // 32
    worst = 0.0
    for n, x in _adversarial_pairs(bits, rng):
        stats = {}
        got = xm.modinv(x, n, nwords, wpl, stats)
        if math.gcd(x, n) == 1:
            assert got == pow(x, -1, n)
        else:
            assert got is None
        assert stats["worst_row_sum"] <= 1 << 30
        worst = max(worst, stats["worst_de_in_B"])
    assert worst < 2 * (xm.rounds_for(nwords) + 1) + 1          # far below the 2^31 the signed overflow word holds


def test_lane_step_growth_is_additive_for_the_worst_matrices():
    """DS::apply on lane values sitting AT the bound, with the matrices of largest row sums and alternating signs, and the
    largest multiple of n: one round adds at most B + 1 (2B + 1 with the multiple of n) to the largest lane value."""
    rng = random.Random(1)
    for wpl in (1, 2, 4):
        B = 1 << (32 * wpl)
        N = [B - 1] * xm.LANES
        for t in (0, 1, 50, 787):
            M = (t + 1) * (B + 1) - 1
            pats = [[M] * xm.LANES, [-M] * xm.LANES, [M if L % 2 else -M for L in range(xm.LANES)],
                    [rng.choice((-M, M, M - 1, 0)) for _ in range(xm.LANES)]]
            mats = [(1 << 30, 0), (-(1 << 30), 0), (1 << 29, -(1 << 29)), (-(1 << 29), 1 << 29), ((1 << 30) - 1, -1), (1, -((1 << 30) - 1)),
                    (-(1 << 29) - 5, (1 << 29) - 5)]
            for X in pats:
                for Y in pats:
                    for a, b in mats:
                        out = xm.apply(a, X, b, Y, 0, N, wpl)
                        assert max(abs(v) for v in out) <= M + B + 1
                        out = xm.apply(a, X, b, Y, -((1 << 30) - 1), N, wpl)
                        assert max(abs(v) for v in out) <= M + 2 * B + 1


def test_divsteps_matrices_have_bounded_rows_and_track_the_low_words():
    """|u| + |v| <= 2^30, |q| + |r| <= 2^30 for any low words, and the matrix really maps (f, g) to multiples of 2^30."""
    rng = random.Random(7)
    for _ in range(3000):
        f, g = rng.getrandbits(32) | 1, rng.getrandbits(32)
        eta0 = rng.choice((-1, -1, -5, 3, 0, 17, -29))
        eta, u, v, q, r = xm.divsteps30(eta0, f, g)
        assert abs(u) + abs(v) <= 1 << 30 and abs(q) + abs(r) <= 1 << 30
        assert (u * f + v * g) % (1 << 30) == 0 and (q * f + r * g) % (1 << 30) == 0
        assert (u * r - v * q) in (1 << 30, -(1 << 30))            # determinant +-2^30: the steps are invertible over the odd f
