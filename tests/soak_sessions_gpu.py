"""Soak of the session coalescer on the GPU (TEST INFRASTRUCTURE; run by hand on the box: python tests/soak_sessions_gpu.py SECONDS SEED).

Round after round: a random number of concurrent perform_secure_comparison sessions (1 .. 600) on one Initiator / KeyHolder pair, random
key size (1024-bit / l = 16 or 2048-bit / l = 32), inputs that mix plaintexts, ciphertexts, equal and adjacent values and negatives;
every result must decrypt to x <= y, and every few rounds the same sessions run again one by one WITHOUT the coalescer under the same
per-session random streams and every message of every session must be equal (tests/_coalesce_harness.py)."""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _coalesce_harness import run_sessions  # noqa: E402
from conftest import oracle_dgk, oracle_paillier  # noqa: E402

from protocols.secure_comparison_amd import DGK, Paillier  # noqa: E402
from protocols.secure_comparison_amd.schemes import default_engine  # noqa: E402


def main() -> None:
    seconds, seed = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    keys = json.load(open(os.path.join(ROOT, "tests", "golden", "keys.json")))
    eng = default_engine()
    worlds = []
    for pbits, dname, l in ((1024, "dgk_1024_l16", 16), (2048, "dgk_2048_l32", 32)):
        sk, d = oracle_paillier(keys, pbits), oracle_dgk(keys, dname)
        bob_p = Paillier(sk.n, sk.p, sk.q, engine=eng)
        bob_d = DGK(d.n, d.g, d.h, d.u, d.t, d.p, d.q, d.v_p, d.v_q, engine=eng, randomizer_bits=400)
        worlds.append((l, bob_p, bob_d, bob_p.public_copy()))
    t0, rounds, sessions, compared = time.time(), 0, 0, 0
    while time.time() - t0 < seconds:
        l, bob_p, bob_d, pub = rng.choice(worlds)
        n = rng.choice((1, 2, 3, 17, 64, 65, 200, 600))
        top = 1 << l
        vals = []
        for i in range(n):
            x = rng.randrange(top)
            y = x if i % 5 == 0 else (min(top - 1, x + 1) if i % 5 == 1 else rng.randrange(top))
            if i % 11 == 0:
                x, y = -rng.randrange(1, 1000), rng.choice((-5, 0, 5))
            vals.append((x, y))
        pairs = [((pub.unsafe_encrypt(x), y) if i % 7 == 3 else ((pub.unsafe_encrypt(x), pub.unsafe_encrypt(y)) if i % 7 == 5 else (x, y))) for i, (x, y) in enumerate(vals)]
        check = rounds % 4 == 0 and n <= 200
        res, sent, stats = run_sessions(pairs, l, bob_p, bob_d, coalesce=True, alice_paillier=pub, strict=False, seed=seed * 1000 + rounds, replay=check)
        dec = eng.download(bob_p.decrypt_raw_batch(eng.upload(res, 2 * bob_p.mod_n.nwords)))
        want = [int(x <= y) for x, y in vals]
        if dec != want:
            raise SystemExit(f"round {rounds}: {sum(a != b for a, b in zip(dec, want))} of {n} results decrypt wrong (seed {seed})")
        if stats["alice"]["largest"] != n or stats["alice"]["fallbacks"] or stats["bob"]["fallbacks"]:
            raise SystemExit(f"round {rounds}: the sessions did not run as one batch: {stats}")
        if check:
            pairs2 = [((pub.unsafe_encrypt(x), y) if i % 7 == 3 else ((pub.unsafe_encrypt(x), pub.unsafe_encrypt(y)) if i % 7 == 5 else (x, y))) for i, (x, y) in enumerate(vals)]
            res2, sent2, _ = run_sessions(pairs2, l, bob_p, bob_d, coalesce=False, alice_paillier=pub, strict=False, seed=seed * 1000 + rounds)
            if res2 != res or sent2 != sent:
                bad = [k for k in sent if sent[k] != sent2.get(k)]
                raise SystemExit(f"round {rounds}: coalesced and uncoalesced runs differ in {len(bad)} messages, e.g. {bad[:3]} (seed {seed})")
            compared += n
        rounds += 1
        sessions += n
        if rounds % 10 == 0:
            print(f"{rounds} rounds ok ({sessions} sessions decrypt right, {compared} compared message by message)", flush=True)
    bob_p.shut_down()
    print(f"session soak finished: {rounds} rounds, {sessions} concurrent single comparisons decrypt to [x <= y]; {compared} of them equal the uncoalesced "
          f"path message by message; seed {seed}")


if __name__ == "__main__":
    main()
