"""Dev tool: pair exponentiation rate (x^e mod p^2, 3B items) for exponents that are all squarings vs a normal exponent,
per kernel policy -- separates the cost of the pair squarings from that of the pair products."""
import json, os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from protocols.secure_comparison_amd.schemes import default_engine

keys = json.load(open(bench.KEYS))
p = int(keys["paillier_2048"]["p"], 16)
eng = default_engine()
rng = random.Random(1)
count = int(sys.argv[1]) if len(sys.argv) > 1 else 196608
m1, m2 = eng.modulus(p), eng.modulus(p * p, 64)
x = eng.upload([rng.randrange(p) for _ in range(256)], 32).repeat((count // 256, 1)).contiguous()
for mode in (0, 2):
    eng.set_onelane_mode(mode)
    for name, e in (("2^1023 (squarings only)", 1 << 1023), ("p (sliding window)", p), ("all ones (most products)", (1 << 1024) - 1)):
        eng.modexp_shared_sq(m1, m2, x, e); torch.cuda.synchronize(); eng.mac_counter(reset=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.modexp_shared_sq(m1, m2, x, e); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"onelane_mode {mode}  {name:28s} {ms:8.2f} ms  {eng.mac_counter()/ms/1e9:6.2f} T", flush=True)
    for name, e in (("k_vm 2^1023", 1 << 1023), ("k_vm p", p)):
        eng.modexp_shared(m1, x, e); torch.cuda.synchronize(); eng.mac_counter(reset=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.modexp_shared(m1, x, e); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"onelane_mode {mode}  {name:28s} {ms:8.2f} ms  {eng.mac_counter()/ms/1e9:6.2f} T", flush=True)
