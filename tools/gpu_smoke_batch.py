"""End-to-end batch vs oracle on the GPU (dev tool)."""
import json, os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from protocols.secure_comparison_amd import Paillier, DGK
from protocols.secure_comparison_amd.batch import BatchDraws, BatchTrace, secure_comparison_batch
from protocols.secure_comparison_amd.schemes import default_engine
from oracle import sc_oracle as o

K = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "keys.json")))
def run(pbits, dname, B, use_crt, rbits=400):
    rng = random.Random(99)
    pkj = K[f"paillier_{pbits}"]; p, q = int(pkj["p"], 16), int(pkj["q"], 16)
    dj = K[dname]; l = dj["l"]
    osk = o.PaillierKey(p * q, p, q)
    od = o.DGKKey(int(dj["p"], 16) * int(dj["q"], 16), int(dj["g"], 16), int(dj["h"], 16), int(dj["u"], 16), dj["t"], int(dj["p"], 16), int(dj["q"], 16), int(dj["v_p"], 16), int(dj["v_q"], 16))
    eng = default_engine()
    bob_p = Paillier(p * q, p, q, use_crt=use_crt); alice_p = bob_p.public_copy()
    bob_d = DGK(od.n, od.g, od.h, od.u, od.t, od.p, od.q, od.v_p, od.v_q, randomizer_bits=rbits); alice_d = bob_d.public_copy()
    xs = [rng.randrange(1 << l) for _ in range(B)]; ys = [rng.randrange(1 << l) for _ in range(B)]
    for i in range(0, B, 4): ys[i] = xs[i]
    for i in range(1, B, 8): ys[i] = max(0, xs[i] - 1)
    drs = [o.draw(rng, l, osk, od, rbits) for _ in range(B)]
    x_enc = [osk.randomize(osk.enc_raw(x), rng.randrange(1, osk.n)) for x in xs]
    y_enc = [osk.randomize(osk.enc_raw(y), rng.randrange(1, osk.n)) for y in ys]
    t0 = time.time()
    traces = [dict() for _ in range(B)]
    exp = [o.compare(xe, ye, l, osk, od, dr, True, tr) for xe, ye, dr, tr in zip(x_enc, y_enc, drs, traces)]
    t_cpu = time.time() - t0
    nw = alice_p.mod_n.nwords; ew = (od.u.bit_length() + 31) // 32; er = (rbits + 31) // 32
    up = eng.upload
    def bitmajor(rows, words):  # rows[b][i] -> [l+1][B][words]
        return torch.stack([up([rows[b][i] for b in range(B)], words) for i in range(l + 1)])
    draws = BatchDraws(
        r=up([d.r for d in drs], nw), delta_a=eng.upload_u64([d.delta_a for d in drs]),
        rhos=bitmajor([d.rhos for d in drs], ew),
        permutation=torch.tensor([d.perm for d in drs], dtype=torch.int64, device=eng.device),
        rho_z=up([d.rho_z for d in drs], nw),
        r_bob_dgk=bitmajor([[d.r_d] + d.r_beta for d in drs], er), r_alice_dgk=bitmajor([d.r_c for d in drs], er),
        rho_zeta_1=up([d.rho_zeta1 for d in drs], nw), rho_zeta_2=up([d.rho_zeta2 for d in drs], nw), rho_delta_b=up([d.rho_delta_b for d in drs], nw))
    # oracle randomizes c_i AFTER the shuffle with r_c[k] for output position k: reorder exponents to pre-shuffle order
    inv_rc = [[None] * (l + 1) for _ in range(B)]
    for b, d in enumerate(drs):
        for k, src in enumerate(d.perm): inv_rc[b][src] = d.r_c[k]
    draws.r_alice_dgk = bitmajor(inv_rc, er)
    tr = BatchTrace()
    tx, ty = up(x_enc, 2 * nw), up(y_enc, 2 * nw)
    torch.cuda.synchronize(); t0 = time.time()
    res = secure_comparison_batch(tx, ty, l, alice_p, alice_d, bob_p, bob_d, draws, True, tr)
    torch.cuda.synchronize(); t_gpu = time.time() - t0
    got = eng.download(res)
    bad = sum(1 for a, b in zip(got, exp) if a != b)
    dec = [osk.dec_raw(g) for g in got]
    wrong = sum(1 for d, x, y in zip(dec, xs, ys) if d != int(x <= y))
    zbad = sum(1 for a, t in zip(eng.download(tr.z_enc), traces) if a != t["z_enc"])
    cbad = sum(1 for b in range(B) for i in range(l + 1) if eng.download(tr.c_sent[i, b:b+1])[0] != traces[b]["c_enc"][i]) if B <= 64 else -1
    print(f"paillier {pbits} {dname} B={B} crt={use_crt}: result mismatches {bad}, wrong decryptions {wrong}, z_enc mismatches {zbad}, c_sent mismatches {cbad}; cpu oracle {t_cpu:.2f}s, gpu {t_gpu:.3f}s", flush=True)
    return bad == 0 and wrong == 0

ok = True
ok &= run(1024, "dgk_1024_l16", 48, False)
ok &= run(1024, "dgk_1024_l16", 48, True)
ok &= run(2048, "dgk_2048_l32", 32, True)
ok &= run(2048, "dgk_2048_l32", 32, False)
print("ALL OK" if ok else "FAILURES")
