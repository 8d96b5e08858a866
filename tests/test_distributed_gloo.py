"""N > 1 path on the CPU: two gloo ranks shard a batch, run every step on their block (test-only OracleEngine) and
all-gather the results; the reassembled batch must equal the unsharded run."""
import os
import random
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, oracle_dgk, oracle_paillier


def _worker(rank, world, port, B, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import json

    from _oracle_engine import OracleEngine
    from oracle import sc_oracle as o
    from protocols.secure_comparison_amd import DGK, Paillier
    from protocols.secure_comparison_amd.batch import secure_comparison_batch
    from protocols.secure_comparison_amd.distributed import all_gather_results, shard_bounds
    from test_host_logic_cpu import make_draws

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    keys = json.load(open(os.path.join(ROOT, "tests", "golden", "keys.json")))
    osk, od = oracle_paillier(keys, 1024), oracle_dgk(keys, "dgk_tiny_l16")
    eng = OracleEngine()
    bob_p = Paillier(osk.n, osk.p, osk.q, engine=eng)
    bob_d = DGK(od.n, od.g, od.h, od.u, od.t, od.p, od.q, od.v_p, od.v_q, engine=eng, randomizer_bits=50)
    rng = random.Random(5)  # every rank builds the same global batch, then takes its block
    L = 16
    xs = [rng.randrange(1 << L) for _ in range(B)]
    ys = [xs[i] if i % 4 == 0 else rng.randrange(1 << L) for i in range(B)]
    drs = [o.draw(rng, L, osk, od, 50) for _ in range(B)]
    x_enc, y_enc = [osk.enc_raw(x) for x in xs], [osk.enc_raw(y) for y in ys]
    lo, hi = shard_bounds(B, rank, world)
    nw = bob_p.mod_n.nwords
    draws = make_draws(eng, drs[lo:hi], L, nw, (od.u.bit_length() + 31) // 32, 2)
    local = secure_comparison_batch(eng.upload(x_enc[lo:hi], 2 * nw), eng.upload(y_enc[lo:hi], 2 * nw), L, bob_p.public_copy(),
                                    bob_d.public_copy(), bob_p, bob_d, draws)
    full = all_gather_results(local, B)
    from protocols.secure_comparison_amd.distributed import all_gather_planes

    if B % world == 0:   # the blocked wire-batch layout [rank][plane][b][words]
        mine = torch.arange((L + 1) * (hi - lo) * 3, dtype=torch.int32).reshape(L + 1, hi - lo, 3) + 100000 * rank
        blocked = all_gather_planes(mine)
        assert blocked.shape == (world, L + 1, hi - lo, 3)
        for r in range(world):
            assert int(blocked[r, 0, 0, 0]) == 100000 * r and int(blocked[r, L, hi - lo - 1, 2]) == 100000 * r + (L + 1) * (hi - lo) * 3 - 1
    if rank == 0:
        expect = [o.compare(a, b, L, osk, od, d, True) for a, b, d in zip(x_enc, y_enc, drs)]
        ret["ok"] = eng.download(full) == expect and [osk.dec_raw(v) for v in expect] == [int(x <= y) for x, y in zip(xs, ys)]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [6, 5])  # even and ragged shards
def test_two_rank_sharding(B):
    ret = mp.get_context("spawn").Manager().dict()
    port = 29500 + random.randrange(2000)
    mp.spawn(_worker, args=(2, port, B, ret), nprocs=2, join=True)
    assert ret.get("ok") is True


def test_shard_bounds():
    from protocols.secure_comparison_amd.distributed import shard_bounds

    for total in (0, 1, 7, 64, 65536):
        for world in (1, 2, 3, 8):
            blocks = [shard_bounds(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            assert max(h - l for l, h in blocks) - min(h - l for l, h in blocks) <= 1
