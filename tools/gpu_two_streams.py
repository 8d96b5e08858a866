"""Experiment: one batch split into two halves driven by two host threads on two HIP streams (two library contexts), so that
the latency-bound launches of one half (inversion trees, xgcd) overlap with the wide launches of the other."""
import json, os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import bench
from protocols.secure_comparison_amd import DGK, Paillier
from protocols.secure_comparison_amd.batch import secure_comparison_batch
from protocols.secure_comparison_amd.engine import Engine

B, l, rbits = 65536, 32, 400
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
SIZES = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [B // NS] * NS
keys = json.load(open(bench.KEYS))
pj, dj = keys["paillier_2048"], keys["dgk_2048_l32"]
p, q = int(pj["p"], 16), int(pj["q"], 16)
H = lambda k: int(dj[k], 16)
parts = []
for i in range(NS):
    eng = Engine()
    bob_p = Paillier(p * q, p, q, engine=eng); alice_p = bob_p.public_copy()
    bob_d = DGK(H("p") * H("q"), H("g"), H("h"), H("u"), dj["t"], H("p"), H("q"), H("v_p"), H("v_q"), engine=eng, randomizer_bits=rbits, fixed_base_window=20)
    alice_d = bob_d.public_copy(); bob_d.prepare(), alice_d.prepare()
    x, y, x_enc, y_enc, draws = bench.synth_inputs(eng, l, alice_p, bob_p, bob_d, SIZES[i], rbits, i)
    parts.append(dict(eng=eng, ap=alice_p, ad=alice_d, bp=bob_p, bd=bob_d, x=x, y=y, xe=x_enc, ye=y_enc, dr=draws, stream=torch.cuda.Stream()))
torch.cuda.synchronize()

def work(pt, out, k):
    with torch.cuda.stream(pt["stream"]):
        out[k] = secure_comparison_batch(pt["xe"], pt["ye"], l, pt["ap"], pt["ad"], pt["bp"], pt["bd"], pt["dr"], randomize=True)

for rep in range(4):
    out = [None] * NS
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(pt, out, k)) for k, pt in enumerate(parts)]
    [t.start() for t in ths]; [t.join() for t in ths]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"rep {rep}: streams {SIZES}: {dt*1e3:.1f} ms -> {sum(SIZES)/dt:.0f} cmp/s", flush=True)
for k, pt in enumerate(parts):
    dec = pt["bp"].decrypt_raw_batch(out[k])
    assert bool((dec[:, 0] == (pt["x"] <= pt["y"]).to(torch.int32)).all().item())
print("results correct")
