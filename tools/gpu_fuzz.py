"""Randomised cross-check of the C ABI against Python integers over all kernel configurations (dev tool, GPU).

usage: python tools/gpu_fuzz.py [seconds] [seed]
Draws modulus sizes from 40 to 8300 bits (every (G, L) configuration incl. the small-batch ones), operands with adversarial limb
patterns (all-ones 29-bit limbs, powers of two, n - k) and random ones, and compares products, exponentiations (shared,
per-element, pair form), inversions and wide-operand reductions with pow() / int arithmetic.  `run()` raises AssertionError on the
first mismatch; tests/test_gpu_fuzz.py runs a short budget of it."""
import math
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))

SIZES = [40, 64, 127, 256, 500, 514, 1024, 1030, 1536, 1550, 1600, 2048, 2060, 3072, 3100, 3200, 4096, 4150, 6144, 6200, 6400, 8192, 8300]
# the largest moduli each configuration accepts (capacity 29 G L minus the 8 guard bits) and their neighbours
CAPS = [522, 1044, 1566, 1624, 2088, 3132, 3248, 4176, 6264, 6496, 8352]
SIZES += [c - d for c in CAPS for d in (8, 9, 31, 33)]
SIZES += [1000, 1020, 1024, 1024, 1027, 1028, 1029]   # around the one-lane configuration's capacity (28 * 37 - 8 bits)


def pattern(rng, bits):
    k = rng.randrange(6)
    if k == 0:
        return (1 << bits) - 1 - 2 * rng.randrange(1 << 20)          # all-ones limbs
    if k == 1:
        return (1 << (bits - 1)) + 1 + 2 * rng.randrange(1 << 20)    # sparse
    if k == 2:                                                       # all ones except a few random holes
        v = (1 << bits) - 1
        for _ in range(rng.randrange(4)):
            v ^= 1 << rng.randrange(1, bits - 1)
        return v | 1 | (1 << (bits - 1))
    return rng.getrandbits(bits) | (1 << (bits - 1)) | 1


def one_round(eng, rng):
    bits = rng.choice(SIZES) if rng.random() < 0.7 else rng.randrange(40, 8300)
    n = pattern(rng, bits)
    eng.set_latency_mode(rng.choice([0, 1, 2]))
    eng.set_onelane_mode(rng.choice([0, 1, 2, 2]))      # the one-lane (1, 37) kernels forced on for the moduli they fit
    try:
        mod = eng.modulus(n)
    except Exception as exc:                                         # larger than the largest configuration
        assert bits > 8300 - 40, (bits, exc)
        return bits
    B = rng.choice([1, 3, 17, 64, 100])
    vals = [n - 1, n - 2, 1, 0, (1 << (bits - 1)) - 1, (1 << (bits - 2)) + 12345, n >> 1]
    vals = (vals + [rng.randrange(n) for _ in range(B)])[:max(B, 7)]
    rev = list(reversed(vals))
    t, u = eng.upload(vals, mod.nwords), eng.upload(rev, mod.nwords)
    assert eng.download(eng.modmul(mod, t, u)) == [a * b % n for a, b in zip(vals, rev)], ("modmul", bits, n)
    ebits = rng.choice([1, 2, 17, 64, 160, 400, min(bits, 1100)])
    e = rng.getrandbits(ebits) | (1 << (ebits - 1))
    if rng.random() < 0.2:
        e = (1 << ebits) - 1
    assert eng.download(eng.modexp_shared(mod, t, e, mul_into=u)) == [pow(a, e, n) * b % n for a, b in zip(vals, rev)], ("modexp", bits, n, e)
    vb = rng.choice([1, 5, 35, 67])
    ev = [rng.getrandbits(vb) for _ in vals]
    got = eng.download(eng.modexp_var(mod, t, eng.upload(ev, (vb + 31) // 32), vb))
    assert got == [pow(a, x, n) for a, x in zip(vals, ev)], ("modexp_var", bits, n)
    wide = [rng.getrandbits(2 * 32 * mod.nwords) for _ in range(5)] + [(1 << (2 * 32 * mod.nwords)) - 1]
    assert eng.download(eng.modexp_shared(mod, eng.upload(wide, 2 * mod.nwords), 5)) == [pow(w, 5, n) for w in wide], ("wide", bits, n)
    inv_in = [v for v in vals if v and math.gcd(v, n) == 1] * rng.choice([1, 1, 9])
    if inv_in:
        assert eng.download(eng.modinv(mod, eng.upload(inv_in, mod.nwords))) == [pow(v, -1, n) for v in inv_in], ("modinv", bits, n)
    if bits <= 4100 and eng.supports_sq(mod):
        try:
            mod2 = eng.modulus(n * n, 2 * mod.nwords)
        except Exception:
            mod2 = None
        if mod2 is not None:
            assert eng.download(eng.modexp_shared_sq(mod, mod2, t, e)) == [pow(a, e, n * n) for a in vals], ("pair", bits, n, e)
    return bits


def run(budget: float, seed: int, eng=None, verbose: bool = True) -> int:
    """Fuzz for `budget` seconds; returns the number of rounds completed."""
    if eng is None:
        from protocols.secure_comparison_amd.engine import Engine

        eng = Engine()
    rng = random.Random(seed)
    t_end = time.time() + budget
    rounds = 0
    while time.time() < t_end:
        bits = one_round(eng, rng)
        rounds += 1
        if verbose and rounds % 20 == 0:
            print(f"{rounds} rounds ok ({bits} bits last)", flush=True)
    return rounds


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    print(f"fuzz finished: {run(budget, seed)} rounds, seed {seed}, no mismatch")
