"""Rank body for tests/test_launcher_cpu.py: started by protocols.secure_comparison_amd.launcher.spawn_ranks (the code path
`bench.py --gpus N` takes), it joins a gloo group from the launcher's environment, runs its shard of a small batch with the
test-only OracleEngine injected into the product classes, proves the collective sees every rank (all-reduce of ones) and
all-gathers the results; rank 0 writes what it saw to the JSON file named on the command line."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main() -> None:
    out_path, asked, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    fail_rank = int(sys.argv[4]) if len(sys.argv) > 4 else -1
    import torch
    import torch.distributed as dist

    from _oracle_engine import OracleEngine
    from conftest import oracle_dgk, oracle_paillier
    from oracle import sc_oracle as o
    from protocols.secure_comparison_amd import DGK, Paillier, launcher
    from protocols.secure_comparison_amd.batch import secure_comparison_batch
    from protocols.secure_comparison_amd.distributed import all_gather_results, shard_bounds
    from test_host_logic_cpu import make_draws

    rank, local_rank, world = launcher.expect_world(asked)
    if rank == fail_rank:
        sys.exit(7)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    keys = json.load(open(os.path.join(ROOT, "tests", "golden", "keys.json")))
    osk, od = oracle_paillier(keys, 1024), oracle_dgk(keys, "dgk_tiny_l16")
    eng = OracleEngine()
    bob_p = Paillier(osk.n, osk.p, osk.q, engine=eng)
    bob_d = DGK(od.n, od.g, od.h, od.u, od.t, od.p, od.q, od.v_p, od.v_q, engine=eng, randomizer_bits=50)
    rng = random.Random(11)
    L = 16
    xs = [rng.randrange(1 << L) for _ in range(B)]
    ys = [xs[i] if i % 3 == 0 else rng.randrange(1 << L) for i in range(B)]
    drs = [o.draw(rng, L, osk, od, 50) for _ in range(B)]
    x_enc, y_enc = [osk.enc_raw(x) for x in xs], [osk.enc_raw(y) for y in ys]
    lo, hi = shard_bounds(B, rank, world)
    nw = bob_p.mod_n.nwords
    draws = make_draws(eng, drs[lo:hi], L, nw, (od.u.bit_length() + 31) // 32, 2)
    local = secure_comparison_batch(eng.upload(x_enc[lo:hi], 2 * nw), eng.upload(y_enc[lo:hi], 2 * nw), L, bob_p.public_copy(),
                                    bob_d.public_copy(), bob_p, bob_d, draws)
    ones = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(ones)
    full = all_gather_results(local, B)
    if rank == 0:
        expect = [o.compare(a, b, L, osk, od, d, True) for a, b, d in zip(x_enc, y_enc, drs)]
        json.dump({"ranks_seen": int(ones.item()), "world": world, "equal": eng.download(full) == expect,
                   "decrypts": [osk.dec_raw(v) for v in expect] == [int(x <= y) for x, y in zip(xs, ys)]}, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
