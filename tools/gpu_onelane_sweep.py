"""Dev tool: x^e mod p (1024-bit modulus and exponent) for a sweep of batch sizes on the two-lane kernel k_vm<2,18> and on the
one-lane kernel k_vm<1,37,28> -- the data behind the one-lane policy (sc_lib.hip::onelane_for)."""
import json, os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from protocols.secure_comparison_amd.schemes import default_engine

keys = json.load(open(bench.KEYS))
p = int(keys["paillier_2048"]["p"], 16)
eng = default_engine()
rng = random.Random(1)
m1 = eng.modulus(p)
base = eng.upload([rng.randrange(p) for _ in range(4096)], 32)
for count in [int(a) for a in sys.argv[1:]] or [32768, 58368, 65536, 98304, 106496, 131072, 163840, 196608, 212992, 262144, 327680, 393216]:
    x = base.repeat((count // 4096, 1)).contiguous()
    row = []
    for mode in (0, 2, 1):
        eng.set_onelane_mode(mode)
        eng.modexp_shared(m1, x, p); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); eng.modexp_shared(m1, x, p); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        row.append(best)
    print(f"{count:8d} items ({count / 131072:5.2f} one-lane rounds)   two-lane {row[0]:7.2f} ms   one-lane {row[1]:7.2f} ms   policy {row[2]:7.2f} ms", flush=True)
eng.set_onelane_mode(1)
