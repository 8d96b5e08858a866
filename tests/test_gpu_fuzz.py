"""A short budget of the randomised C-ABI cross-check (tools/gpu_fuzz.py): random modulus sizes over every kernel configuration
and both small-batch policies, adversarial limb patterns, every primitive compared with Python integers."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_randomised_primitives_against_python_ints(engine):
    import gpu_fuzz

    try:
        rounds = gpu_fuzz.run(25.0, seed=20261003, eng=engine, verbose=False)
    finally:
        engine.set_latency_mode(0)
    assert rounds >= 10


# ------------------------------------------------------------------------------------------ whole comparisons on random keys
def protocol_round(engine, rng):
    """One batch of whole comparisons on freshly generated keys of random sizes, against oracle.compare (bit for bit)."""
    import torch  # noqa: F401

    from oracle import sc_oracle as o
    from protocols.secure_comparison_amd import keygen
    from protocols.secure_comparison_amd.batch import secure_comparison_batch
    from protocols.secure_comparison_amd import DGK, Paillier
    from test_gpu_parity import _draw_tensors

    l = rng.choice([1, 3, 8, 16, 24])
    v_bits = rng.choice([24, 40, 64])
    n_bits = rng.choice([b for b in (192, 256, 384, 512, 768, 1024) if b // 2 >= l + v_bits + 16])
    pbits = rng.choice([128, 192, 256, 384, 512, 768, 1024, 1100])
    u = keygen.next_prime(1 << (l + 2))
    k = keygen.dgk_key(v_bits, n_bits, u)
    dgk = o.DGKKey(k["n"], k["g"], k["h"], k["u"], k["t"], k["p"], k["q"], k["v_p"], k["v_q"])
    p, q = keygen.paillier_primes(pbits)
    sk = o.PaillierKey(p * q, p, q)
    rbits = int(2.5 * v_bits)
    use_crt = rng.random() < 0.7
    engine.set_latency_mode(rng.choice([0, 1, 2]))
    engine.set_onelane_mode(rng.choice([0, 1, 2]))
    bob_p = Paillier(sk.n, sk.p, sk.q, engine=engine, use_crt=use_crt, use_pairs=rng.random() < 0.8)
    bob_d = DGK(dgk.n, dgk.g, dgk.h, dgk.u, dgk.t, dgk.p, dgk.q, dgk.v_p, dgk.v_q, engine=engine, randomizer_bits=rbits,
                fixed_base_window=rng.choice([1, 3, 8, 11]), use_crt=rng.random() < 0.7)
    alice_p, alice_d = bob_p.public_copy(), bob_d.public_copy()
    B = rng.choice([1, 2, 7, 33])
    xs = [rng.randrange(1 << l) for _ in range(B)]
    ys = [xs[i] if i % 3 == 0 else rng.randrange(1 << l) for i in range(B)]
    shuffle = rng.random() < 0.5                                    # a batch shuffles either every comparison or none
    drs = [o.draw(rng, l, sk, dgk, rbits, shuffle=shuffle) for _ in range(B)]
    x_enc = [sk.randomize(sk.enc_raw(x), 1 + rng.randrange(sk.n - 1)) for x in xs]
    y_enc = [sk.randomize(sk.enc_raw(y), 1 + rng.randrange(sk.n - 1)) for y in ys]
    nw = bob_p.mod_n.nwords
    draws = _draw_tensors(engine, drs, l, nw, (dgk.u.bit_length() + 31) // 32, (rbits + 31) // 32, engine.device)
    got = engine.download(secure_comparison_batch(engine.upload(x_enc, 2 * nw), engine.upload(y_enc, 2 * nw), l, alice_p, alice_d,
                                                  bob_p, bob_d, draws))
    expect = [o.compare(a, b, l, sk, dgk, d, True) for a, b, d in zip(x_enc, y_enc, drs)]
    assert got == expect, ("protocol", l, v_bits, n_bits, pbits, use_crt, B)
    assert [sk.dec_raw(c) for c in got] == [int(x <= y) for x, y in zip(xs, ys)]


def run_protocol_fuzz(engine, budget, seed):
    import random
    import time

    rng = random.Random(seed)
    t_end, rounds = time.time() + budget, 0
    try:
        while time.time() < t_end:
            protocol_round(engine, rng)
            rounds += 1
            if rounds % 10 == 0:
                print(rounds, "protocol rounds ok", flush=True)
    finally:
        engine.set_latency_mode(0)
        engine.set_onelane_mode(1)
    return rounds


def test_randomised_whole_comparisons_on_random_keys(engine):
    """Fresh Paillier (128 ... 1100-bit) and DGK (192 ... 1024-bit) keys, l from 1 to 24, CRT on / off, shuffles on / off, every
    small-batch policy: each batch bit-exact against the oracle and decrypting to [x <= y]."""
    assert run_protocol_fuzz(engine, 30.0, seed=31337) >= 3


if __name__ == "__main__":  # python tests/test_gpu_fuzz.py <seconds> <seed>: a longer run of the protocol fuzz
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from protocols.secure_comparison_amd.schemes import default_engine

    secs, sd = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1
    print("protocol fuzz:", run_protocol_fuzz(default_engine(), secs, sd), "rounds, seed", sd, "no mismatch")
