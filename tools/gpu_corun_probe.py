"""Dev tool: how long an under-filling launch takes alone and when it is placed on a chip that already runs another context's
launch (wave placement), or is placed first and joined by the other launch afterwards.

The launch: the key holder's first CRT stage (rho mod q)^(p mod q-1) mod q for 3 x 4096 numbers on a context that shares the chip
(768 waves of k_vm<4,9>, one wave per workgroup).  A trailing y / n says whether the other context was still busy when the launch
ended (i.e. whether the two really overlapped)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import bench
from protocols.secure_comparison_amd import DGK, Paillier
from protocols.secure_comparison_amd.engine import Engine

B, l, rbits = 4096, 16, 400
keys = json.load(open(bench.KEYS))
pj, dj = keys["paillier_2048"], keys["dgk_2048_l16"]
p, q = int(pj["p"], 16), int(pj["q"], 16)
H = lambda k: int(dj[k], 16)  # noqa: E731
engs = [Engine(), Engine()]
for e in engs:
    e.set_chip_share(2)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
A, Bn = engs
bob_p = Paillier(p * q, p, q, engine=A); alice_p = bob_p.public_copy()
bob_d = DGK(H("p") * H("q"), H("g"), H("h"), H("u"), dj["t"], H("p"), H("q"), H("v_p"), H("v_q"), engine=A, randomizer_bits=rbits, fixed_base_window=16)
x, y, x_enc, y_enc, draws = bench.synth_inputs(A, l, alice_p, bob_p, bob_d, B, rbits, 0)
alice_p2 = Paillier(p * q, engine=Bn)
hw = (q.bit_length() + 31) // 32
m_q, m_q2 = A.modulus(q, hw), Bn.modulus(q, hw)
rho3 = torch.cat([draws.rho_zeta_1, draws.rho_zeta_2, draws.rho_delta_b], dim=0).contiguous()
rho24 = torch.cat([rho3] * 16, dim=0).contiguous()                 # 196608 numbers: chip-filling for ~17 ms
big_rho = torch.cat([draws.rho_z] * 16, dim=0).contiguous()        # 65536 numbers: a chip-filling pair launch (~105 ms, 256 VGPRs)
e_small = p % (q - 1)


def launch_a():
    with torch.cuda.stream(sa):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(sa); A.modexp_shared(m_q, rho3, e_small); e1.record(sa)
    return e0, e1


def run(name, before=None, after=None, pause=0.0, reps=5, delay=0.0015):
    ts, ov = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        if pause:
            time.sleep(pause)
        eb = torch.cuda.Event()
        if before:
            with torch.cuda.stream(sb):
                before(); eb.record(sb)
            t_end = time.perf_counter() + delay
            while time.perf_counter() < t_end:
                pass
        e0, e1 = launch_a()
        if after:
            with torch.cuda.stream(sb):
                after(); eb.record(sb)
        sa.synchronize()
        ov.append("y" if (before or after) and not eb.query() else "n")
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{name:84s} " + " ".join(f"{t:6.2f}{o}" for t, o in zip(ts, ov)) + " ms", flush=True)


with torch.cuda.stream(sa):
    A.modexp_shared(m_q, rho3, e_small)
with torch.cuda.stream(sb):
    alice_p2.randomizer_batch(big_rho); Bn.modexp_shared(m_q2, rho24, e_small); Bn.modexp_shared(m_q2, rho3, e_small)
torch.cuda.synchronize()
run("alone, back to back")
run("alone, after 20 ms of idle", pause=0.02)
run("placed 1.5 ms after a chip-filling pair launch (256 VGPRs x 2 waves) of another context", before=lambda: alice_p2.randomizer_batch(big_rho))
run("placed 1.5 ms after a chip-filling modexp launch of another context", before=lambda: Bn.modexp_shared(m_q2, rho24, e_small))
run("placed just BEFORE the same chip-filling modexp launch of another context", after=lambda: Bn.modexp_shared(m_q2, rho24, e_small))
run("placed 1.5 ms after the same under-filling launch of another context", before=lambda: Bn.modexp_shared(m_q2, rho3, e_small))
run("placed just before the same under-filling launch of another context", after=lambda: Bn.modexp_shared(m_q2, rho3, e_small))

# the situation of a configs[1] step: the other context's blinding launch (17 x 4096 items of k_vm<4,18>, ~1.7 ms, two waves on
# every SIMD but registers and LDS to spare) arrives shortly before / after this launch
from protocols.secure_comparison_amd import Initiator
alice_d2 = DGK(H("p") * H("q"), H("g"), H("h"), H("u"), dj["t"], engine=Bn, randomizer_bits=rbits, fixed_base_window=16)
alice_d2.prepare()
with torch.cuda.stream(sb):
    c17 = alice_d2.randomize_batch(None, draws.r_alice_dgk.reshape((l + 1) * B, -1)).reshape(l + 1, B, -1)
    hr = alice_d2.randomize_batch(None, draws.r_alice_dgk.reshape((l + 1) * B, -1)).reshape(l + 1, B, -1)
    blind = lambda: Bn.initiator_step4i(alice_d2.key, l, c17, draws.rhos, None, hr, ready=True)   # noqa: E731
    blind()
torch.cuda.synchronize()
for d in (0.0, 0.0003, 0.0006, 0.001, 0.0015):
    run(f"placed {d * 1e3:.1f} ms after the other context's blinding launch (k_vm<4,18>, 1.7 ms)", before=blind, delay=d)
run("placed just before the other context's blinding launch", after=blind)

# a sustained stretch of under-filling launches with no host synchronisation in between: does the chip slow down when only 768
# waves are resident for tens of milliseconds?
torch.cuda.synchronize()
with torch.cuda.stream(sa):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(13)]
    evs[0].record(sa)
    for i in range(12):
        A.modexp_shared(m_q, rho3, e_small); evs[i + 1].record(sa)
torch.cuda.synchronize()
print("twelve launches queued back to back, ms each: " + " ".join(f"{evs[i].elapsed_time(evs[i + 1]):5.2f}" for i in range(12)), flush=True)
# the same after a chip-filling launch of the SAME context and stream (as after the blinding launch of a step)
with torch.cuda.stream(sa):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
    A.modexp_shared(m_q, rho24, e_small)
    evs[0].record(sa)
    for i in range(7):
        A.modexp_shared(m_q, rho3, e_small); evs[i + 1].record(sa)
torch.cuda.synchronize()
print("seven launches after a chip-filling one, ms each:     " + " ".join(f"{evs[i].elapsed_time(evs[i + 1]):5.2f}" for i in range(7)), flush=True)
