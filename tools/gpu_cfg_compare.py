"""Dev tool: modexp throughput at 1536 / 3072 / 6144 bits (direct and pair form) for the library named by SC_AMD_LIB."""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from protocols.secure_comparison_amd.engine import Engine
eng = Engine(); rng = random.Random(1)
print("lib:", os.environ.get("SC_AMD_LIB", "default"))
for bits, B in ((1536, 65536), (3072, 32768), (6144, 16384)):
    n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    e = rng.getrandbits(512) | (1 << 511) | 1
    mod = eng.modulus(n)
    xs = [rng.randrange(n) for _ in range(64)]
    x = eng.upload(xs, mod.nwords).repeat((B // 64, 1)).contiguous()
    out = eng.modexp_shared(mod, x, e); torch.cuda.synchronize()
    got = eng.download(out[:64])
    assert got == [pow(v, e, n) for v in xs], "direct modexp mismatch"
    eng.mac_counter(reset=True)
    t0 = time.perf_counter(); eng.modexp_shared(mod, x, e); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    macs = eng.mac_counter()
    print(f"direct {bits:5d}-bit B={B:6d}: {dt*1e3:8.2f} ms  executed {macs/dt/1e12:6.2f} T limb-MAC/s", flush=True)
for bits, B in ((1536, 32768), (3072, 16384)):
    n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    e = rng.getrandbits(512) | (1 << 511) | 1
    mod = eng.modulus(n); mod2 = eng.modulus(n * n)
    xs = [rng.randrange(n) for _ in range(64)]
    x = eng.upload(xs, mod.nwords).repeat((B // 64, 1)).contiguous()
    for name, fn in (("direct n^2", lambda: eng.modexp_shared(mod2, x, e)),
                     ("pair   n^2", (lambda: eng.modexp_shared_sq(mod, mod2, x, e)) if eng.supports_sq(mod) else None)):
        if fn is None:
            print(f"{name} {bits}-bit: unsupported"); continue
        out = fn(); torch.cuda.synchronize()
        assert eng.download(out[:64]) == [pow(v, e, n * n) for v in xs], name + " mismatch"
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name} ({bits}-bit n) B={B:6d}: {dt*1e3:8.2f} ms", flush=True)
