"""Calibration point for FETCH_SIZE / WRITE_SIZE on the kernels' own limb-form table accesses (MI355X guide, HBM section:
'other access widths are uncalibrated: calibrate on a known byte count in your own access pattern').

Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and again with `--pmc WRITE_SIZE`; the probe launch is the k_vm<4,18,29>
dispatch with grid 131072 whose known byte counts this script prints (JSON on the last line)."""
import json, os, random, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from protocols.secure_comparison_amd.engine import Engine

B, ENTRIES, READS = 65536, 32, 320
eng = Engine(); rng = random.Random(5)
n = rng.getrandbits(2048) | (1 << 2047) | 1
mod = eng.modulus(n)
xs = [rng.randrange(n) for _ in range(64)]
x = eng.upload(xs, mod.nwords).repeat((B // 64, 1)).contiguous()
for _ in range(2):
    out, S = eng.table_traffic_probe(mod, x, ENTRIES, READS)
    torch.cuda.synchronize()
assert bool((out == x).all().item())
print(json.dumps({"kernel": "sc::k_vm<4, 18, 29>", "items": B, "row_bytes": S * 4,
                  "known_read_bytes": B * (READS * S * 4 + mod.nwords * 4), "known_write_bytes": B * (ENTRIES * S * 4 + mod.nwords * 4),
                  "table_footprint_bytes": 2048 * 16 * ENTRIES * S * 4}))
