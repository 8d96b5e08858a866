"""Quick GPU check of the primitive ops against Python ints (dev tool; the real tests are in tests/)."""
import json, os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from protocols.secure_comparison_amd.engine import Engine

rng = random.Random(5)
eng = Engine()
keys = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "keys.json")))
def check(name, got, exp):
    bad = sum(1 for a, b in zip(got, exp) if a != b)
    print(f"{name}: {len(exp)} items, {bad} mismatches", flush=True)
    return bad == 0

ok = True
for bits in (128, 1024, 2048, 4096):
    n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    mod = eng.modulus(n)
    B = 200
    a = [rng.randrange(n) for _ in range(B)]; b = [rng.randrange(n) for _ in range(B)]
    a[0] = 0; a[1] = 1; a[2] = n - 1; b[2] = n - 1
    ta, tb = eng.upload(a, mod.nwords), eng.upload(b, mod.nwords)
    ok &= check(f"modmul {bits}", eng.download(eng.modmul(mod, ta, tb)), [x * y % n for x, y in zip(a, b)])
    ok &= check(f"modmul bcast {bits}", eng.download(eng.modmul(mod, ta, tb[:1])), [x * b[0] % n for x in a])
    ok &= check(f"modmul_const {bits}", eng.download(eng.modmul_const(mod, ta, b[5])), [x * b[5] % n for x in a])
    e = rng.getrandbits(bits // 2) | 1
    ok &= check(f"modexp_shared {bits}", eng.download(eng.modexp_shared(mod, ta, e)), [pow(x, e, n) for x in a])
    ok &= check(f"modexp_shared mulinto {bits}", eng.download(eng.modexp_shared(mod, ta, e, mul_into=tb)), [pow(x, e, n) * y % n for x, y in zip(a, b)])
    for ee in (0, 1, 2, 3, 65537):
        ok &= check(f"modexp_shared e={ee} {bits}", eng.download(eng.modexp_shared(mod, ta, ee)), [pow(x, ee, n) for x in a])
    ebits = 35
    ev = [rng.getrandbits(ebits) for _ in range(B)]; ev[0] = 0; ev[1] = 1; ev[2] = (1 << ebits) - 1
    te = eng.upload(ev, 2)
    ok &= check(f"modexp_var {bits}", eng.download(eng.modexp_var(mod, ta, te, ebits)), [pow(x, y, n) for x, y in zip(a, ev)])
    h = rng.randrange(2, n)
    for win in (4, 8):
        fb = eng.fixed_base(mod, h, 100, win)
        rv = [rng.getrandbits(100) for _ in range(B)]; rv[0] = 0; rv[1] = 1
        tr = eng.upload(rv, 4)
        ok &= check(f"fixedbase w{win} {bits}", eng.download(eng.fixedbase_pow(fb, tr)), [pow(h, y, n) for y in rv])
        ok &= check(f"fixedbase mulinto w{win} {bits}", eng.download(eng.fixedbase_pow(fb, tr, mul_into=ta)), [pow(h, y, n) * x % n for x, y in zip(a, rv)])
        ok &= check(f"modexp_var+fb w{win} {bits}", eng.download(eng.modexp_var(mod, ta, te, ebits, fb, tr)), [pow(x, y, n) * pow(h, z, n) % n for x, y, z in zip(a, ev, rv)])
print("ALL OK" if ok else "FAILURES")
