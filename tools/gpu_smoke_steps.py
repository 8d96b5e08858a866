"""Quick GPU check of inversion / Paillier / DGK step kernels against the oracle (dev tool)."""
import json, os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from protocols.secure_comparison_amd.engine import Engine, NotInvertibleError
from oracle import sc_oracle as o

rng = random.Random(11)
eng = Engine()
K = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "keys.json")))
def check(name, got, exp):
    bad = sum(1 for a, b in zip(got, exp) if a != b)
    print(f"{name}: {len(exp)} items, {bad} mismatches", flush=True)
    return bad == 0 and len(got) == len(exp)
ok = True
def pk(bits):
    k = K[f"paillier_{bits}"]; p, q = int(k["p"], 16), int(k["q"], 16); return o.PaillierKey(p * q, p, q)
def dk(name):
    k = K[name]; p, q = int(k["p"], 16), int(k["q"], 16)
    return o.DGKKey(p * q, int(k["g"], 16), int(k["h"], 16), int(k["u"], 16), k["t"], p, q, int(k["v_p"], 16), int(k["v_q"], 16))

# ---- modinv
for bits, B in ((128, 5), (1024, 40), (2048, 3000), (4096, 700)):
    n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    mod = eng.modulus(n)
    a = []
    while len(a) < B:
        x = rng.randrange(1, n)
        try: pow(x, -1, n); a.append(x)
        except ValueError: pass
    a[0] = 1; a[-1] = n - 1
    t0 = time.time(); got = eng.download(eng.modinv(mod, eng.upload(a, mod.nwords))); dt = time.time() - t0
    ok &= check(f"modinv {bits} ({dt*1e3:.1f} ms)", got, [pow(x, -1, n) for x in a])
# non-invertible detection
n = 3 * 5 * 7 * 11 * 13 * (rng.getrandbits(1000) | 1)
mod = eng.modulus(n)
try:
    eng.modinv(mod, eng.upload([2, 15, 4], mod.nwords)); print("non-invertible NOT detected"); ok = False
except NotInvertibleError as e:
    print("non-invertible detected:", e)

# ---- Paillier
for bits in (1024, 2048):
    key = pk(bits); n, n2 = key.n, key.n2
    mN, mN2 = eng.modulus(n), eng.modulus(n2)
    B = 100
    ms = [rng.randrange(n) for _ in range(B)]; ms[0] = 0; ms[1] = 1; ms[2] = n - 1
    ok &= check(f"paillier enc_raw {bits}", eng.download(eng.paillier_encrypt_raw(mN2, n, eng.upload(ms, mN.nwords))), [key.enc_raw(m) for m in ms])
    big = [m + (1 << 32) for m in ms]  # plaintexts above N (2^l + r)
    ok &= check(f"paillier enc_raw wide {bits}", eng.download(eng.paillier_encrypt_raw(mN2, n, eng.upload(big, mN.nwords + 1))), [key.enc_raw(m) for m in big])
    cts = [key.randomize(key.enc_raw(m), rng.randrange(1, n)) for m in ms]
    tc = eng.upload(cts, mN2.nwords)
    x = eng.modexp_shared(mN2, tc, key.lam)
    ok &= check(f"paillier decrypt {bits}", eng.download(eng.paillier_l_mul(mN, key.mu, x)), ms)
    rhos = [rng.randrange(1, n) for _ in range(B)]
    ok &= check(f"paillier randomize {bits}", eng.download(eng.modexp_shared(mN2, eng.upload(rhos, mN2.nwords), n, mul_into=tc)), [key.randomize(c, r) for c, r in zip(cts, rhos)])
    for l in (16, 32, 64):
        rs = [rng.randrange(n) for _ in range(B)]; rs[0] = 0; rs[1] = n - 1; rs[2] = (n - 1) // 2; rs[3] = (n - 1) // 2 - 1
        m1, al, at, rsm, rsh = eng.plain_alice(eng.upload(rs, mN.nwords), n, l)
        M = (1 << 64) - 1
        ok &= check(f"alice m1 l={l}", eng.download(m1), [(1 << l) + r for r in rs])
        ok &= check(f"alice alpha l={l}", [v & M for v in al.tolist()], [r % (1 << l) for r in rs])
        ok &= check(f"alice alpha_t l={l}", [v & M for v in at.tolist()], [(r - n) % (1 << l) for r in rs])
        ok &= check(f"alice rsmall l={l}", rsm.tolist(), [int(r < (n - 1) // 2) for r in rs])
        ok &= check(f"alice rshift l={l}", eng.download(rsh), [r >> l for r in rs])
        be, db, z1, z2 = eng.plain_bob(eng.upload(rs, mN.nwords), n, l)
        ok &= check(f"bob beta l={l}", [v & M for v in be.tolist()], [r % (1 << l) for r in rs])
        ok &= check(f"bob dbit l={l}", db.tolist(), [int(r < (n - 1) // 2) for r in rs])
        ok &= check(f"bob zeta1 l={l}", eng.download(z1), [r >> l for r in rs])
        ok &= check(f"bob zeta2 l={l}", eng.download(z2), [((r + n) >> l) if r < (n - 1) // 2 else (r >> l) for r in rs])

# ---- DGK zero test + step 4
for name in ("dgk_tiny_l16", "dgk_1024_l16", "dgk_2048_l32"):
    d = dk(name); l = K[name]["l"]
    mn, mp = eng.modulus(d.n), eng.modulus(d.p)
    B = 64
    ms = [rng.randrange(d.u) for _ in range(B)]
    for i in range(0, B, 3): ms[i] = 0
    cts = [d.randomize(d.enc_raw(m), rng.getrandbits(100)) for m in ms]
    fl = eng.modexp_shared_isone(mp, eng.upload(cts, mn.nwords), d.v_p)
    ok &= check(f"is_zero {name}", fl.tolist(), [int(d.is_zero(c)) for c in cts])
    # step 4 against the oracle
    pkey = pk(1024 if "2048" not in name else 2048)
    alphas, atils, rsm, das, exp_c, betas, binv, ds, dinv = [], [], [], [], [], [], [], [], []
    for _ in range(B):
        r = rng.randrange(pkey.n); da = rng.randrange(2)
        z = rng.randrange(pkey.n)
        alpha = o.step_3(r, l)
        d_enc = d.randomize(o.step_4a(z, d, pkey, l), rng.getrandbits(100))
        b_enc = [d.randomize(c, rng.getrandbits(100)) for c in o.step_4b(z % (1 << l), l, d)]
        d2 = o.step_4c(d_enc, r, d, pkey)
        xor = o.step_4d(alpha, b_enc, d)
        w, at = o.step_4e(r, alpha, xor, d2, pkey, d)
        w = o.step_4f(w, d)
        s, _ = o.step_4g(da)
        exp_c.append(o.step_4h(s, alpha, at, d2, b_enc, w, da, d))
        alphas.append(o.from_bits(alpha)); atils.append(o.from_bits(at)); rsm.append(int(r < (pkey.n - 1) // 2)); das.append(da)
        betas.append(b_enc); ds.append(d_enc)
    nw = mn.nwords
    tb = torch.stack([eng.upload([betas[c][i] for c in range(B)], nw) for i in range(l)])           # [l][B][nw]
    tbi = eng.modinv(mn, tb.reshape(l * B, nw)).reshape(l, B, nw)
    td = eng.upload(ds, nw); tdi = eng.modinv(mn, td)
    out = eng.dgk_step4(mn, d.g, pow(d.g, -1, d.n), l, tb, tbi, td, tdi, eng.upload_u64(alphas), eng.upload_u64(atils),
                        eng.upload_u64(rsm), eng.upload_u64(das))
    got = eng.download(out.reshape((l + 1) * B, nw))
    exp = [exp_c[c][i] for i in range(l + 1) for c in range(B)]
    ok &= check(f"dgk_step4 {name}", got, exp)
print("peak probe MAC/s:", eng.peak_probe())
print("ALL OK" if ok else "FAILURES")
