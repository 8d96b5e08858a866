"""Dev tool: how much of the blinding / fixed-base launches is exposed memory latency?  Same launches with random exponents
(table rows scattered over the 0.6 / 6 GB table) and with all-zero exponents (every item reads the same cached rows)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from protocols.secure_comparison_amd import DGK, Initiator, Paillier
from protocols.secure_comparison_amd.schemes import default_engine

B, l, rbits = 65536, 32, 400
keys = json.load(open(bench.KEYS))
pj, dj = keys["paillier_2048"], keys["dgk_2048_l32"]
p, q = int(pj["p"], 16), int(pj["q"], 16)
H = lambda k: int(dj[k], 16)
eng = default_engine()
bob_p = Paillier(p * q, p, q); alice_p = bob_p.public_copy()
for w in (16, 20):
    bob_d = DGK(H("p") * H("q"), H("g"), H("h"), H("u"), dj["t"], H("p"), H("q"), H("v_p"), H("v_q"), randomizer_bits=rbits, fixed_base_window=w)
    alice_d = bob_d.public_copy()
    x, y, x_enc, y_enc, draws = bench.synth_inputs(eng, l, alice_p, bob_p, bob_d, B, rbits, 0)
    c33 = alice_d.randomize_batch(None, draws.r_alice_dgk.reshape((l + 1) * B, -1))
    e_rand = draws.r_alice_dgk.reshape((l + 1) * B, -1)
    rho = draws.rhos.reshape((l + 1) * B, -1)
    def t(name, fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        print(f"w={w}  {name:52s} {e0.elapsed_time(e1):8.2f} ms", flush=True)
    t("fixed base h^r (Alice, mod n), random exponents", lambda: alice_d.randomize_batch(None, e_rand))
    t("fixed base h^r (Alice, mod n), zero exponents", lambda: alice_d.randomize_batch(None, torch.zeros_like(e_rand)))
    c33p = c33.reshape(l + 1, B, -1)
    t("blind c^rho * h^r, random rho, random r", lambda: Initiator.step_4i_batch(c33p, alice_d, draws.rhos, None, draws.r_alice_dgk))
    t("blind c^rho * h^r, random rho, zero r", lambda: Initiator.step_4i_batch(c33p, alice_d, draws.rhos, None, torch.zeros_like(draws.r_alice_dgk)))
    t("blind c^rho only (no fixed base)", lambda: eng.modexp_var(alice_d.mod_n, c33, rho, 35))
    t("bob h^r CRT (mod p, q tables), random", lambda: bob_d.randomize_batch(None, draws.r_bob_dgk.reshape((l + 1) * B, -1)))
    t("bob h^r CRT, zero exponents", lambda: bob_d.randomize_batch(None, torch.zeros_like(e_rand)))
