"""Dev tool: correctness and time of the top-level inversion kernel (sc_modinv on <= 48 residues = one k_xgcd launch)."""
import math, os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from protocols.secure_comparison_amd.engine import Engine, NotInvertibleError
eng = Engine(); rng = random.Random(4)
for bits in (64, 200, 1024, 2048, 3072, 4096, 6144, 8192):
    n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    mod = eng.modulus(n)
    xs = [1, n - 1, 2] + [rng.randrange(1, n) for _ in range(37)]
    xs = [x for x in xs if math.gcd(x, n) == 1]
    t = eng.upload(xs, mod.nwords)
    got = eng.download(eng.modinv(mod, t))
    ok = got == [pow(x, -1, n) for x in xs]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): eng.modinv(mod, t)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{bits:5d} bits: {len(xs)} residues correct={ok}  {dt*1e3:7.3f} ms per call", flush=True)
n = 3 * 5 * 7 * (rng.getrandbits(1000) | 1)
mod = eng.modulus(n)
try:
    eng.modinv(mod, eng.upload([2, 35, 4], mod.nwords)); print("non-invertible NOT detected")
except NotInvertibleError as e:
    print("non-invertible detected at", e.index)
