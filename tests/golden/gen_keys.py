"""Generate the fixed key fixtures (tests/golden/keys.json) with the repo's own oracle.

Run from the repo root:  python tests/golden/gen_keys.py
Deterministic (seeded); DGK keys follow SC/keyholder.py:161-166 (v_bits=160, u=next_prime(2^(l+2))).
These are TEST keys: the secret parts are public in this file on purpose.
"""
import json
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import sc_oracle as o  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "keys.json")


def main() -> None:
    rng = random.Random(0x5EC0C0DE)
    keys = {}
    for bits in (1024, 2048, 3072):
        k = o.PaillierKey.generate(bits, rng)
        keys[f"paillier_{bits}"] = {"p": hex(k.p), "q": hex(k.q)}
        print("paillier", bits, flush=True)
    specs = [("dgk_tiny_l16", 20, 128, 16), ("dgk_1024_l16", 160, 1024, 16), ("dgk_2048_l16", 160, 2048, 16),
             ("dgk_2048_l32", 160, 2048, 32), ("dgk_2048_l64", 160, 2048, 64), ("dgk_3072_l64", 160, 3072, 64)]
    for name, v_bits, n_bits, l in specs:
        u = o.next_prime(1 << (l + 2))
        k = o.DGKKey.generate(v_bits, n_bits, u, rng)
        keys[name] = {"p": hex(k.p), "q": hex(k.q), "v_p": hex(k.v_p), "v_q": hex(k.v_q), "g": hex(k.g),
                      "h": hex(k.h), "u": hex(k.u), "t": v_bits, "l": l}
        print(name, flush=True)
    with open(OUT, "w") as f:
        json.dump(keys, f, indent=1)


if __name__ == "__main__":
    main()
