"""Dev tool: executed multiply-add rate of the hot launch groups of a batch step, one by one, for one or more builds of the
library (A/B on the same box):

    python tools/gpu_kernel_rates.py [lib.so ...]        # default: the in-tree library

Each build runs in its own child process (SC_AMD_LIB).  Shapes are those of BASELINE configs[2] (B = 65536, l = 32,
2048-bit keys): Alice's rho^N (k_pvm<4,18>), the key holder's CRT halves (k_vm<1,37> one-lane -- k_vm<2,18> below 196608 items -- then k_pvm<2,18> over 3B items),
the zero tests (k_vm<1,37>, 33B items), blinding + re-randomization (k_vm<4,18>, 33B items), decrypt (k_pvm<2,18>).
"""
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def child() -> None:
    import torch

    import bench
    from protocols.secure_comparison_amd import DGK, Paillier
    from protocols.secure_comparison_amd.schemes import default_engine

    B = int(os.environ.get("KR_BATCH", "65536"))
    l, rbits = 32, 400
    keys = json.load(open(bench.KEYS))
    pj, dj = keys["paillier_2048"], keys["dgk_2048_l32"]
    p, q = int(pj["p"], 16), int(pj["q"], 16)
    H = lambda k: int(dj[k], 16)  # noqa: E731
    eng = default_engine()
    if os.environ.get("KR_ONELANE"):
        eng.set_onelane_mode(int(os.environ["KR_ONELANE"]))
    bob_p = Paillier(p * q, p, q)
    alice_p = bob_p.public_copy()
    bob_d = DGK(H("p") * H("q"), H("g"), H("h"), H("u"), dj["t"], H("p"), H("q"), H("v_p"), H("v_q"), randomizer_bits=rbits, fixed_base_window=int(os.environ.get("KR_WINDOW", bench.DEFAULT_FB_WINDOW)))
    alice_d = bob_d.public_copy()
    x, y, x_enc, y_enc, draws = bench.synth_inputs(eng, l, alice_p, bob_p, bob_d, B, rbits, 0)
    from protocols.secure_comparison_amd import Initiator

    hw = (max(p.bit_length(), q.bit_length()) + 31) // 32
    m_p, m_p2, e_small = eng.modulus(p, hw), eng.modulus(p * p, 2 * hw), q % (p - 1)   # the primitives behind the key holder's CRT
    rho3 = torch.cat([draws.rho_zeta_1, draws.rho_zeta_2, draws.rho_delta_b], dim=0)
    nwd = alice_d.mod_n.nwords
    c33 = alice_d.randomize_batch(None, draws.r_alice_dgk.reshape((l + 1) * B, -1))
    c33p = c33.reshape(l + 1, B, nwd)
    y_p = eng.modexp_shared(m_p, rho3, e_small)
    one_lane = 3 * B >= 196608
    out = {}

    def rate(name, fn, reps=2):
        fn()
        torch.cuda.synchronize()
        eng.mac_counter(reset=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[name] = {"ms": round(ms, 3), "T_mac_s": round(eng.mac_counter() / reps / ms / 1e9, 3)}

    rate("alice rho^N mod N^2  (k_pvm<4,18>+k_vm<8,18>, B)", lambda: alice_p.randomize_batch(x_enc, draws.rho_z))
    rate("bob (rho mod p)^e mod p  (%s, 3B)" % ("k_vm<1,37>" if one_lane else "k_vm<2,18>"), lambda: eng.modexp_shared(m_p, rho3, e_small))
    rate("bob y^p mod p^2  (k_pvm<2,18>, 3B)", lambda: eng.modexp_shared_sq(m_p, m_p2, y_p, p))
    rate("bob 3 randomizers, whole CRT path (3B)", lambda: bob_p.randomize_batch(None, rho3))
    rate("bob decrypt, whole CRT path (B)", lambda: bob_p.decrypt_raw_batch(x_enc))
    rate("zero tests + delta_B (k_vm<1,37>, 33B)", lambda: bob_d.any_zero_batch(c33p))
    rate("blind + rerandomize, 4i (k_vm<4,18>, 33B)", lambda: Initiator.step_4i_batch(c33p, alice_d, draws.rhos, None, draws.r_alice_dgk))
    rate("bob g^b h^r, CRT (k_vm<2,18>, 33B)", lambda: bob_d.randomize_batch(None, draws.r_bob_dgk.reshape((l + 1) * B, -1)))
    bits33 = (torch.arange((l + 1) * B, device=c33.device) % 3 == 0).to(torch.uint8)
    rate("bob encrypt bits g^b h^r, CRT (33B)", lambda: bob_d.encrypt_bits_randomized_batch(bits33, draws.r_bob_dgk.reshape((l + 1) * B, -1)))
    flags = [(torch.arange(B, device=c33.device, dtype=torch.int64) * k) & ((1 << l) - 1) for k in (0x9E3779B1, 0x85EBCA77)]
    one_bit = [(torch.arange(B, device=c33.device, dtype=torch.int64) >> k) & 1 for k in (1, 2)]
    rate("steps 4c-4h with their inversions (k_vm<4,18>, B)", lambda: eng.initiator_step4(alice_d.key, l, c33p[0], c33p[1:], flags[0], flags[1], one_bit[0], one_bit[1]))
    rate("inversions mod n (33B)", lambda: alice_d.neg_batch(c33))
    rate("inversion mod N^2 (B)", lambda: alice_p.neg_batch(x_enc))
    print(json.dumps(out))


def main() -> None:
    if os.environ.get("KR_CHILD"):
        child()
        return
    libs = sys.argv[1:] or [""]
    res = {}
    for rep in range(2):
        for lib in libs:
            env = dict(os.environ, KR_CHILD="1")
            if lib:
                env["SC_AMD_LIB"] = os.path.abspath(lib)
            cp = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            if cp.returncode != 0:
                print(lib, "FAILED", cp.stderr[-1500:], flush=True)
                continue
            d = json.loads(cp.stdout.strip().splitlines()[-1])
            for k, v in d.items():
                cur = res.setdefault(k, {}).setdefault(lib or "in-tree", v)
                if v["ms"] < cur["ms"]:
                    res[k][lib or "in-tree"] = v
    names = [lb or "in-tree" for lb in libs]
    print("%-46s" % "launch group" + "".join("%26s" % os.path.basename(n)[-24:] for n in names))
    for k, row in res.items():
        print("%-46s" % k + "".join("%16.2f ms %6.2f T" % (row[n]["ms"], row[n]["T_mac_s"]) if n in row else "%26s" % "-" for n in names), flush=True)


if __name__ == "__main__":
    main()
