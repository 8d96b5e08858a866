"""Montgomery-product rate per kernel configuration (dev tool): times sc_modexp_shared with a 512-bit exponent."""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from protocols.secure_comparison_amd.engine import Engine
eng = Engine(); rng = random.Random(1)
e = rng.getrandbits(512) | (1 << 511) | 1
for bits, B in ((512, 131072), (1024, 65536), (1536, 32768), (2048, 32768), (3072, 16384), (4096, 16384), (6144, 8192), (8192, 4096)):
    n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    mod = eng.modulus(n)
    x = eng.upload([rng.randrange(n) for _ in range(64)], mod.nwords).repeat((B // 64, 1)).contiguous()
    eng.modexp_shared(mod, x, e); torch.cuda.synchronize()
    eng.mac_counter(reset=True)
    t0 = time.perf_counter(); eng.modexp_shared(mod, x, e); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    macs = eng.mac_counter()
    print(f"{bits:5d}-bit B={B:6d}: {dt*1e3:8.2f} ms  executed {macs/dt/1e12:6.2f} T limb-MAC/s", flush=True)
