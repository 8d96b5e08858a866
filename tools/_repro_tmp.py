import json, os, sys
sys.path.insert(0, "/root/repo")
import torch, bench
from protocols.secure_comparison_amd import DGK, Paillier
from protocols.secure_comparison_amd.batch import secure_comparison_batch
from protocols.secure_comparison_amd.engine import Engine
B, seed, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
l, rbits = 32, 400
keys = json.load(open(bench.KEYS)); pj, dj = keys["paillier_2048"], keys["dgk_2048_l32"]
p, q = int(pj["p"], 16), int(pj["q"], 16); H = lambda k: int(dj[k], 16)
eng = Engine(); eng.set_latency_mode(mode)
bob_p = Paillier(p * q, p, q, engine=eng); alice_p = bob_p.public_copy()
bob_d = DGK(H("p") * H("q"), H("g"), H("h"), H("u"), dj["t"], H("p"), H("q"), H("v_p"), H("v_q"), engine=eng, randomizer_bits=rbits, fixed_base_window=20)
alice_d = bob_d.public_copy()
x, y, xe, ye, dr = bench.synth_inputs(eng, l, alice_p, bob_p, bob_d, B, rbits, seed)
try:
    out = secure_comparison_batch(xe, ye, l, alice_p, alice_d, bob_p, bob_d, dr, randomize=True)
    dec = bob_p.decrypt_raw_batch(out)
    print(B, seed, mode, "ok", bool((dec[:, 0] == (x <= y).to(torch.int32)).all().item()))
except Exception as e:
    print(B, seed, mode, "FAIL", e)
