"""Soak run of the headline configuration (not collected by pytest: `python tests/soak_gpu.py <seconds> <seed> [B [pbits dgk_key_name l]]`).

BASELINE configs[2] as bench.py runs it -- B = 65536, l = 32, 2048/2048-bit keys, two concurrent shards with shared window-16
tables, every randomization and the step-4i shuffle -- over and over with FRESH inputs and draws per batch (another seed each
time, generated on the device): every batch must decrypt to [x <= y] in all rows, and a few rows per batch (first, last, both
sides of the shard cut and two random ones) must equal oracle.compare bit for bit.  A progress line per ten batches; the last
line is the total.  Only meaningful on a GPU box; test infrastructure like everything else under tests/."""
from __future__ import annotations

import os
import random
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main() -> int:
    import json

    import bench
    from conftest import GOLDEN, oracle_dgk, oracle_paillier
    from protocols.secure_comparison_amd import DGK, Paillier
    from protocols.secure_comparison_amd.batch import ConcurrentShards, PartySet, split_draws
    from protocols.secure_comparison_amd.distributed import shard_bounds
    from protocols.secure_comparison_amd.engine import Engine
    from test_gpu_round2 import _oracle_rows

    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
    keys = json.load(open(os.path.join(GOLDEN, "keys.json")))
    pbits = int(sys.argv[4]) if len(sys.argv) > 4 else 2048                  # other BASELINE shapes: e.g. 32768 3072 dgk_3072_l64 64
    dname = sys.argv[5] if len(sys.argv) > 5 else "dgk_2048_l32"
    l = int(sys.argv[6]) if len(sys.argv) > 6 else 32
    sk, dgk = oracle_paillier(keys, pbits), oracle_dgk(keys, dname)
    rbits, window = 400, bench.DEFAULT_FB_WINDOW
    engines = [Engine(), Engine()]
    sets = []
    for i, e in enumerate(engines):
        bob_p = Paillier(sk.n, sk.p, sk.q, engine=e)
        bob_d = DGK(dgk.n, dgk.g, dgk.h, dgk.u, dgk.t, dgk.p, dgk.q, dgk.v_p, dgk.v_q, engine=e, randomizer_bits=rbits, fixed_base_window=window)
        alice_d = bob_d.public_copy()
        if i > 0:
            bob_d.share_tables_from(sets[0].bob_dgk)
            alice_d.share_tables_from(sets[0].alice_dgk)
        alice_d.prepare(), bob_d.prepare()
        sets.append(PartySet(bob_p.public_copy(), alice_d, bob_p, bob_d, torch.cuda.Stream()))
    p0 = sets[0]
    bounds = [shard_bounds(B, i, 2) for i in range(2)]
    cut = bounds[0][1]
    runner = ConcurrentShards(sets)
    rng = random.Random(seed0)
    t_end, batches, rows_checked = time.time() + secs, 0, 0
    try:
        while time.time() < t_end:
            seed = seed0 * 100003 + batches
            x, y, x_enc, y_enc, draws = bench.synth_inputs(engines[0], l, p0.alice_paillier, p0.bob_paillier, p0.bob_dgk, B, rbits, seed=seed, shuffle=True)
            shard_inputs = [(x_enc[a:b].contiguous(), y_enc[a:b].contiguous(), d) for (a, b), d in zip(bounds, split_draws(draws, bounds))]
            torch.cuda.synchronize()
            res = torch.cat(runner.run(shard_inputs, l, randomize=True), dim=0)
            dec = p0.bob_paillier.decrypt_raw_batch(res)
            ok = bool(((dec[:, 0] == (x <= y).to(torch.int32)) & (dec[:, 1:] == 0).all(dim=1)).all().item())
            if not ok:
                print(f"MISMATCH: batch {batches} (seed {seed}): a row does not decrypt to [x <= y]", flush=True)
                return 1
            idx = sorted({0, cut - 1, cut, B - 1, rng.randrange(B), rng.randrange(B)})
            if engines[0].download(res[torch.tensor(idx, device=res.device)]) != _oracle_rows(engines[0], idx, l, sk, dgk, x_enc, y_enc, draws):
                print(f"MISMATCH: batch {batches} (seed {seed}): rows {idx} differ from the oracle", flush=True)
                return 1
            batches += 1
            rows_checked += len(idx)
            if batches % 10 == 0:
                print(f"{batches} batches ok ({batches * B} comparisons decrypt right, {rows_checked} rows bit-exact)", flush=True)
    finally:
        runner.close()
        for e in engines:
            e.close()
    print(f"soak finished ({pbits}-bit Paillier, {dname}, l = {l}): {batches} batches of {B} = {batches * B} comparisons decrypt to [x <= y]; {rows_checked} sampled rows equal the oracle bit for bit; seed {seed0}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
