"""Pair-arithmetic exponentiation x^e mod m^2 vs Python pow (dev tool)."""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from protocols.secure_comparison_amd.engine import Engine
eng = Engine(); rng = random.Random(3); ok = True
for bits in (256, 512, 1024, 2048):
    m = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    mm, mm2 = eng.modulus(m), eng.modulus(m * m, 2 * ((bits + 31) // 32))
    assert eng.supports_sq(mm)
    B = 100
    for xw_mult, xs in ((1, [rng.randrange(m) for _ in range(B)]), (2, [rng.randrange(m * m) for _ in range(B)]), (4, [rng.getrandbits(4 * bits) for _ in range(B)])):
        xs[0] = 1; xs[1] = 0 if xw_mult == 1 else m; xs[2] = m - 1
        t = eng.upload(xs, xw_mult * mm.nwords)
        cs = [rng.randrange(m * m) for _ in range(B)]
        tc = eng.upload(cs, mm2.nwords)
        for e in (1, 2, 3, 65537, rng.getrandbits(bits) | 1, m):
            got = eng.download(eng.modexp_shared_sq(mm, mm2, t, e))
            exp = [pow(x, e, m * m) for x in xs]
            bad = sum(a != b for a, b in zip(got, exp)); ok &= bad == 0
            if bad: print(f"bits {bits} xw {xw_mult} e {e.bit_length()}b: {bad} mismatches", flush=True)
        got = eng.download(eng.modexp_shared_sq(mm, mm2, t, m, mul_into=tc))
        bad = sum(a != b for a, b in zip(got, [pow(x, m, m * m) * c % (m * m) for x, c in zip(xs, cs)])); ok &= bad == 0
        if bad: print(f"bits {bits} xw {xw_mult} mulinto: {bad} mismatches")
    print("bits", bits, "done", flush=True)
# speed: N = 2048-bit
bits = 2048
m = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
mm, mm2 = eng.modulus(m), eng.modulus(m * m, 128)
B = 65536
x = eng.upload([rng.randrange(m) for _ in range(64)], 64).repeat((B // 64, 1)).contiguous()
for name, fn in (("pairs", lambda: eng.modexp_shared_sq(mm, mm2, x, m)), ("direct", lambda: eng.modexp_shared(mm2, torch.nn.functional.pad(x, (0, 64)), m))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    print(f"rho^N mod N^2, B={B}, {name}: {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True)
print("ALL OK" if ok else "FAILURES")
