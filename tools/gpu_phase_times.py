"""Per-phase device time of one batch step (dev tool): same calls as batch.secure_comparison_batch with events."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import bench
from protocols.secure_comparison_amd import DGK, Paillier, Initiator, KeyHolder
from protocols.secure_comparison_amd.schemes import default_engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
fbw = int(sys.argv[2]) if len(sys.argv) > 2 else 13
pbits = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
l = int(sys.argv[4]) if len(sys.argv) > 4 else 32
rbits = 400
keys = json.load(open(bench.KEYS))
pj, dj = keys[f"paillier_{pbits}"], keys[f"dgk_{pbits}_l{l}"]
p, q = int(pj["p"], 16), int(pj["q"], 16)
eng = default_engine()
bob_p = Paillier(p * q, p, q); alice_p = bob_p.public_copy()
bob_d = DGK(int(dj["p"], 16) * int(dj["q"], 16), int(dj["g"], 16), int(dj["h"], 16), int(dj["u"], 16), dj["t"], int(dj["p"], 16), int(dj["q"], 16), int(dj["v_p"], 16), int(dj["v_q"], 16), randomizer_bits=rbits, fixed_base_window=fbw)
alice_d = bob_d.public_copy(); alice_d.prepare()
x, y, x_enc, y_enc, draws = bench.synth_inputs(eng, l, alice_p, bob_p, bob_d, B, rbits, 0)
times, macs = {}, {}
def timed(name, fn):
    torch.cuda.synchronize(); eng.mac_counter(reset=True); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    times[name] = times.get(name, 0) + (time.perf_counter() - t0) * 1e3; macs[name] = macs.get(name, 0) + eng.mac_counter(); return r
for it in range(1 if pbits > 2048 else 2):
    times.clear(); macs.clear()
    z_enc, a_plain = timed("A step1 (inv x, enc, 2 mul)", lambda: Initiator.step_1_batch(x_enc, y_enc, l, alice_p, draws.r))
    z_enc = timed("A randomize z (rho^N mod N^2)", lambda: alice_p.randomize_batch(z_enc, draws.rho_z))
    b_plain = timed("B step2 decrypt (CRT)", lambda: KeyHolder.step_2_batch(z_enc, l, bob_p))
    d_enc, beta_enc = timed("B step4a/4b + randomize (g^b h^r)", lambda: KeyHolder.step_4a_4b_batch(b_plain, l, bob_d, bob_p, draws.r_bob_dgk))
    def a4():
        e = alice_d.engine
        ll, count, nw = beta_enc.shape
        inv = timed("A   modinv beta,d", lambda: alice_d.neg_batch(torch.cat([beta_enc.reshape(ll * count, nw), d_enc], dim=0)))
        beta_inv, d_inv = inv[: ll * count].reshape(ll, count, nw), inv[ll * count:]
        return timed("A   step4 fused kernel", lambda: e.dgk_step4(alice_d.mod_n, alice_d.public_key.g, alice_d.g_inv, ll, beta_enc, beta_inv, d_enc, d_inv, a_plain.alpha, a_plain.alpha_tilde, a_plain.r_small, draws.delta_a))
    c_h = a4()
    c_sent = timed("A step4i blind + randomize", lambda: Initiator.step_4i_batch(c_h, alice_d, draws.rhos, None, draws.r_alice_dgk))
    delta_b = timed("B step4j zero tests", lambda: KeyHolder.step_4j_batch(c_sent, bob_d))
    z1, z2, db = timed("B step5 enc", lambda: KeyHolder.step_5_batch(b_plain, delta_b, bob_p))
    rnd = timed("B randomize x3 (CRT)", lambda: bob_p.randomize_batch(torch.cat([z1, z2, db], dim=0), torch.cat([draws.rho_zeta_1, draws.rho_zeta_2, draws.rho_delta_b], dim=0)))
    z1, z2, db = rnd[:B], rnd[B:2 * B], rnd[2 * B:]
    blta = timed("A step6", lambda: Initiator.step_6_batch(draws.delta_a, db, alice_p))
    res = timed("A step7", lambda: Initiator.step_7_batch(z1, z2, a_plain, l, blta, alice_p))
tot = sum(times.values())
for k, v in times.items(): print(f"{k:40s} {v:8.1f} ms  {100*v/tot:5.1f}%  {macs[k]/v/1e9:6.2f} T limb-MAC/s executed")
print(f"total {tot:.1f} ms -> {B/tot*1e3:.0f} cmp/s")
