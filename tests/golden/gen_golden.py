"""Generate tests/golden/kat_primitives.json and tests/golden/comparisons.json with the repo's own oracle.

    python tests/golden/gen_golden.py            # writes the fixtures (pure Python ints)
    /opt/conda/bin/python3.9 tests/golden/gen_golden.py --check   # recomputes with gmpy2 and compares digests

The fixtures pin the oracle (and through it the GPU path) at ciphertext level across interpreters / big-int
back ends; the plaintext-level pin against the reference's own test vectors is tests/test_oracle_reference_vectors.py.
"""
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import sc_oracle as o  # noqa: E402

REF_PAIRS = [(-400, -383), (-1, 0), (0, 2), (1, 10), (230, 269), (1508, 2408), (3122, 6048), (4250, 7804), (8668, 9015)]


def load_keys():
    k = json.load(open(os.path.join(HERE, "keys.json")))

    def pk(bits):
        p, q = int(k[f"paillier_{bits}"]["p"], 16), int(k[f"paillier_{bits}"]["q"], 16)
        return o.PaillierKey(p * q, p, q)

    def dk(name):
        d = k[name]
        p, q = int(d["p"], 16), int(d["q"], 16)
        return o.DGKKey(p * q, int(d["g"], 16), int(d["h"], 16), int(d["u"], 16), d["t"], p, q, int(d["v_p"], 16), int(d["v_q"], 16)), d["l"]

    return pk, dk


def kat_primitives():
    rng = random.Random(0xA11CE)
    out = []
    for bits in (128, 1024, 2048, 3072, 4096, 6144):
        n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        ops = [0, 1, n - 1, 2, (n - 1) // 2] + [rng.randrange(n) for _ in range(5)]
        a = ops
        b = list(reversed(ops))
        e = rng.getrandbits(min(bits, 512)) | 1
        inv_in = []
        while len(inv_in) < 6:
            x = rng.randrange(1, n)
            try:
                o.mod_inv(x, n)
                inv_in.append(x)
            except (ValueError, ZeroDivisionError):
                pass
        inv_in[0], inv_in[1] = 1, n - 1
        small = [0, 1, 2, 3, (1 << 35) - 1, rng.getrandbits(35), rng.getrandbits(67)]
        out.append({
            "bits": bits, "n": hex(n), "a": [hex(x) for x in a], "b": [hex(x) for x in b],
            "mul": [hex(x * y % n) for x, y in zip(a, b)], "e": hex(e), "pow": [hex(o.pow_mod(x, e, n)) for x in a],
            "inv_in": [hex(x) for x in inv_in], "inv": [hex(o.mod_inv(x, n)) for x in inv_in],
            "small_e": [hex(x) for x in small], "pow_small": [hex(o.pow_mod(a[5], x, n)) for x in small],
        })
    return out


def comparisons():
    pk, dk = load_keys()
    sets = []
    for pbits, dname, pairs, rbits, seed in (
        (1024, "dgk_tiny_l16", REF_PAIRS + [(b, a) for a, b in REF_PAIRS] + [(a, a) for a, _ in REF_PAIRS], 50, 1),
        (1024, "dgk_1024_l16", [(23, 42), (42, 23), (7, 7), (0, 65535), (65535, 0)], 400, 2),
        (2048, "dgk_2048_l32", [(23, 42), (42, 23), (123456789, 123456789), (0, 2 ** 32 - 1), (2 ** 32 - 1, 0), (4000000000, 4000000001),
                                (4000000001, 4000000000), (1, 0)], 400, 3),
    ):
        sk = pk(pbits)
        dgk, l = dk(dname)
        rng = random.Random(seed)
        items = []
        for x, y in pairs:
            dr = o.draw(rng, l, sk, dgk, rbits)
            x_enc = sk.randomize(sk.enc_raw(sk.encode(x)), 1 + rng.randrange(sk.n - 1))
            y_enc = sk.randomize(sk.enc_raw(sk.encode(y)), 1 + rng.randrange(sk.n - 1))
            tr = {}
            res = o.compare(x_enc, y_enc, l, sk, dgk, dr, True, tr)
            assert sk.dec_raw(res) == int(x <= y)
            res_static = o.compare(x_enc, y_enc, l, sk, dgk, dr, False)
            assert sk.dec_raw(res_static) == int(x <= y)
            items.append({
                "x": x, "y": y, "x_enc": hex(x_enc), "y_enc": hex(y_enc),
                "draws": {"r": hex(dr.r), "delta_a": dr.delta_a, "rhos": [hex(v) for v in dr.rhos], "perm": dr.perm,
                          "rho_z": hex(dr.rho_z), "r_d": hex(dr.r_d), "r_beta": [hex(v) for v in dr.r_beta],
                          "r_c": [hex(v) for v in dr.r_c], "rho_zeta1": hex(dr.rho_zeta1), "rho_zeta2": hex(dr.rho_zeta2),
                          "rho_delta_b": hex(dr.rho_delta_b)},
                "z_enc": hex(tr["z_enc"]), "z": hex(tr["z"]), "d_sent": hex(tr["d_sent"]),
                "beta_enc_sha256": hashlib.sha256(",".join(hex(v) for v in tr["beta_enc"]).encode()).hexdigest(),
                "c_h_sha256": hashlib.sha256(",".join(hex(v) for v in tr["c_h"]).encode()).hexdigest(),
                "c_sent": [hex(v) for v in tr["c_enc"]], "delta_b": tr["delta_b"],
                "zeta1": hex(tr["zeta1"]), "zeta2": hex(tr["zeta2"]), "delta_b_enc": hex(tr["delta_b_enc"]),
                "result": hex(res), "result_static": hex(res_static), "expected_bit": int(x <= y),
            })
        sets.append({"paillier_bits": pbits, "dgk": dname, "l": l, "rbits": rbits, "items": items})
    return sets


def digest(obj):
    return hashlib.sha256(json.dumps(obj, sort_keys=True).encode()).hexdigest()


def main():
    kat, cmp_ = kat_primitives(), comparisons()
    if "--check" in sys.argv:
        old_k = json.load(open(os.path.join(HERE, "kat_primitives.json")))
        old_c = json.load(open(os.path.join(HERE, "comparisons.json")))
        assert digest(old_k) == digest(kat) and digest(old_c) == digest(cmp_), "fixtures differ between big-int back ends"
        print("check ok (gmpy2=%s): fixtures reproduce bit for bit" % o._HAVE_GMPY2)
        return
    json.dump(kat, open(os.path.join(HERE, "kat_primitives.json"), "w"), indent=0)
    json.dump(cmp_, open(os.path.join(HERE, "comparisons.json"), "w"), indent=0)
    print("written; gmpy2 =", o._HAVE_GMPY2)


if __name__ == "__main__":
    main()
