"""Dev tool: per-step wall time of consecutive batch steps (looks for clock / allocator drift over a sustained run)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import bench
from protocols.secure_comparison_amd import DGK, Paillier
from protocols.secure_comparison_amd.batch import secure_comparison_batch
from protocols.secure_comparison_amd.schemes import default_engine

B, l, rbits, steps = 65536, 32, 400, int(sys.argv[1]) if len(sys.argv) > 1 else 16
keys = json.load(open(bench.KEYS))
pj, dj = keys["paillier_2048"], keys["dgk_2048_l32"]
p, q = int(pj["p"], 16), int(pj["q"], 16)
eng = default_engine()
bob_p = Paillier(p * q, p, q); alice_p = bob_p.public_copy()
H = lambda k: int(dj[k], 16)
bob_d = DGK(H("p") * H("q"), H("g"), H("h"), H("u"), dj["t"], H("p"), H("q"), H("v_p"), H("v_q"), randomizer_bits=rbits, fixed_base_window=20)
alice_d = bob_d.public_copy(); bob_d.prepare(), alice_d.prepare()
x, y, x_enc, y_enc, draws = bench.synth_inputs(eng, l, alice_p, bob_p, bob_d, B, rbits, 0)
for i in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    secure_comparison_batch(x_enc, y_enc, l, alice_p, alice_d, bob_p, bob_d, draws, randomize=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    peak = eng.peak_probe() / 1e12 if i % 4 == 3 else 0
    print(f"step {i:2d}: {dt*1e3:7.1f} ms  mem {torch.cuda.memory_allocated()/2**30:5.2f} GiB" + (f"  probe {peak:.2f} T MAC/s" if peak else ""), flush=True)
