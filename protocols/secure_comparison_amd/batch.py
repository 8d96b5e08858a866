"""Whole-batch driver: both parties' compute for B independent comparisons, device-resident end to end.

This is the unit the throughput metric is defined on: every step a1-a20 of the hot-path table,
including all 4 + 2(l+1) randomizations of the interactive protocol (SC/initiator.py:69-175,
SC/keyholder.py:70-133), excluding key generation, table build, host RNG and transport.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from ._views import cat_rows
from .initiator import Initiator
from .keyholder import KeyHolder
from .schemes import DGK, Paillier


@dataclass
class BatchDraws:
    """Every random input of B comparisons as device arrays (SURVEY 8(a) note 1)."""

    r: torch.Tensor                    # [B][nw]        Alice's blinding value, 0 <= r < N          (SC/initiator.py:250)
    delta_a: torch.Tensor              # [B] u64        Alice's coin                                (:420)
    rhos: torch.Tensor                 # [l+1][B][ew]   blinding exponents in [1, u)                (:512)
    permutation: torch.Tensor | None   # [B][l+1] i64   shuffle (None = do_shuffle False)           (:223)
    rho_z: torch.Tensor                # [B][nw]        Paillier randomizer base for [[z]]          (:109)
    r_bob_dgk: torch.Tensor            # [l+1][B][er]   DGK randomizer exponents for [d], [beta_i]  (SC/keyholder.py:106-108)
    r_alice_dgk: torch.Tensor          # [l+1][B][er]   DGK randomizer exponents for the sent [c_i] (SC/initiator.py:153-154)
    rho_zeta_1: torch.Tensor           # [B][nw]        Paillier randomizer bases                   (SC/keyholder.py:126-128)
    rho_zeta_2: torch.Tensor
    rho_delta_b: torch.Tensor


@dataclass
class BatchTrace:
    z_enc: torch.Tensor | None = None
    z: torch.Tensor | None = None
    d_enc: torch.Tensor | None = None
    beta_enc: torch.Tensor | None = None
    c_step4h: torch.Tensor | None = None
    c_sent: torch.Tensor | None = None
    delta_b: torch.Tensor | None = None
    zeta_1_enc: torch.Tensor | None = None
    zeta_2_enc: torch.Tensor | None = None
    delta_b_enc: torch.Tensor | None = None


def draw_alice(count: int, l: int, paillier: Paillier, dgk: DGK, source: str = "device", generator=None) -> BatchDraws:
    """Alice's random inputs for `count` comparisons, drawn on the device (randomness.py): r below N (SC/initiator.py:250), the
    coin delta_A (:420), rho_i in [1, u) (:512), the shuffle (:223), and the inputs of her 1 + (l+1) randomizations (:109,
    :153-154; :205-210 scaled by the batch).  The key holder's fields stay None."""
    from .randomness import random_bits, random_coins, random_permutations, uniform_below

    e, n, u = paillier.engine, paillier.public_key.n, dgk.public_key.u
    return BatchDraws(
        r=uniform_below(n, count, e, source, generator), delta_a=random_coins(count, e, source, generator),
        rhos=uniform_below(u, (l + 1) * count, e, source, generator, nonzero=True).reshape(l + 1, count, -1),
        permutation=random_permutations(count, l + 1, e, source, generator),
        rho_z=uniform_below(n, count, e, source, generator, nonzero=True), r_bob_dgk=None,
        r_alice_dgk=random_bits(dgk.randomizer_bits, (l + 1, count), e, source, generator),
        rho_zeta_1=None, rho_zeta_2=None, rho_delta_b=None)


def draw_bob(count: int, l: int, paillier: Paillier, dgk: DGK, source: str = "device", generator=None) -> BatchDraws:
    """The key holder's random inputs: the exponents of his (l+1) DGK randomizations (SC/keyholder.py:106-108) and the bases of
    his 3 Paillier ones (:126-128) -- three row blocks of one array, so that they join again without a copy."""
    from .randomness import random_bits, uniform_below

    e = paillier.engine
    rho = uniform_below(paillier.public_key.n, 3 * count, e, source, generator, nonzero=True)
    return BatchDraws(r=None, delta_a=None, rhos=None, permutation=None, rho_z=None,
                      r_bob_dgk=random_bits(dgk.randomizer_bits, (l + 1, count), e, source, generator), r_alice_dgk=None,
                      rho_zeta_1=rho[:count], rho_zeta_2=rho[count:2 * count], rho_delta_b=rho[2 * count:])


def boot_pools(count: int, l: int, alice_paillier: Paillier, alice_dgk: DGK, bob_paillier: Paillier, bob_dgk: DGK,
               source: str = "device", generator=None) -> None:
    """Pre-generate every randomizer `count` comparisons consume (the batched form of the two players'
    _start_randomness_generation: 1 + (l+1) for Alice, 3 + (l+1) for Bob, per comparison)."""
    alice_paillier.boot_randomness_generation_batch(count, source, generator)
    alice_dgk.boot_randomness_generation_batch((l + 1) * count, source, generator)
    bob_paillier.boot_randomness_generation_batch(3 * count, source, generator)
    bob_dgk.boot_randomness_generation_batch((l + 1) * count, source, generator)


class _AheadOfTime:
    """The 4 + 2(l+1) randomizer exponentiations of a batch -- 60 % of its multiply-adds, and functions of the draws alone --
    queued on a SECOND library context and stream while the protocol's critical path runs on the caller's: the reference
    pre-generates its randomizers in background workers the same way (boot_randomness_generation, SC/initiator.py:205-210,
    SC/keyholder.py:174-179).  Every job records an event; the step that consumes a randomizer waits for it on the main stream
    and applies it with one modular product (SC_STEP_RANDOMIZERS_READY).  Same residues as the fused launches."""

    def __init__(self, side: "PartySet", draws: BatchDraws, l: int) -> None:
        count = draws.r.shape[0]
        jobs = (("rz", lambda: side.alice_paillier.randomizer_batch(draws.rho_z)),                       # needed first
                ("hr_bob", lambda: side.bob_dgk.randomize_batch(None, draws.r_bob_dgk.reshape((l + 1) * count, -1))),
                ("hr_alice", lambda: side.alice_dgk.randomize_batch(None, draws.r_alice_dgk.reshape((l + 1) * count, -1))),
                ("r3", lambda: side.bob_paillier.randomizer_batch(cat_rows([draws.rho_zeta_1, draws.rho_zeta_2, draws.rho_delta_b]))))
        self._out = {}
        self.main = None
        if not draws.r.is_cuda:                    # host tensors (the CPU test tier's stand-in engine): nothing to overlap
            for name, fn in jobs:
                self._out[name] = (fn(), None)
            return
        self.main = torch.cuda.current_stream(draws.r.device)
        with torch.cuda.stream(side.stream):
            side.stream.wait_stream(self.main)     # the draws were produced on the caller's stream
            for name, fn in jobs:
                t = fn()
                t.record_stream(self.main)         # allocated on the side stream, consumed on the main one
                ev = torch.cuda.Event()
                ev.record(side.stream)
                self._out[name] = (t, ev)

    def take(self, name: str) -> torch.Tensor:
        t, ev = self._out.pop(name)
        if ev is not None:
            self.main.wait_event(ev)
        return t


def secure_comparison_batch(x_enc: torch.Tensor, y_enc: torch.Tensor, l: int, alice_paillier: Paillier, alice_dgk: DGK,
                            bob_paillier: Paillier, bob_dgk: DGK, draws: BatchDraws, randomize: bool | str = True,
                            trace: BatchTrace | None = None, side: "PartySet | None" = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """[[x <= y]] for B comparisons.  x_enc, y_enc: [B][2nw] Paillier ciphertexts under Bob's key.
    randomize: True = every `.randomize()` of the interactive protocol, computed from the injected randomizer inputs in `draws`;
    "pool" = the same randomizations with pre-generated randomizers from the schemes' device pools (boot_pools), i.e. the
    online phase of a deployment that generates randomness ahead of time like the reference's background workers;
    False = the static step chain without randomization.
    side: both parties' scheme objects bound to a SECOND engine and stream (same keys): with randomize=True the randomizer
    exponentiations run there, concurrently with the critical path on the caller's stream (see _AheadOfTime) -- inside this
    call, so they are part of the step; identical results.
    out: the [B][2nw] array the results are written to (e.g. a shard's row block of the whole batch's result array)."""
    if randomize == "pool":
        return _secure_comparison_batch_pooled(x_enc, y_enc, l, alice_paillier, alice_dgk, bob_paillier, bob_dgk, draws)
    if side is not None and randomize:
        ahead = _AheadOfTime(side, draws, l)
        count = x_enc.shape[0]
        z_enc, a_plain = Initiator.step_1_batch(x_enc, y_enc, l, alice_paillier, draws.r, None)
        z_enc = alice_paillier.add_batch(z_enc, ahead.take("rz"))
        b_plain, d_enc, beta_enc = KeyHolder.step_2_4b_batch(z_enc, l, bob_paillier, bob_dgk,
                                                             ahead.take("hr_bob").reshape(l + 1, count, -1), randomizers_ready=True)
        c_sent, c_h = Initiator.step_4_batch(d_enc, beta_enc, a_plain, draws.delta_a, alice_dgk, draws.rhos, draws.permutation,
                                             ahead.take("hr_alice").reshape(l + 1, count, -1), want_unblinded=trace is not None,
                                             randomizers_ready=True)
        delta_b, zeta_1_enc, zeta_2_enc, delta_b_enc = KeyHolder.step_4j_5_batch(c_sent, b_plain, bob_paillier, bob_dgk, ahead.take("r3"),
                                                                                 randomizers_ready=True)
        result = Initiator.step_6_7_batch(draws.delta_a, delta_b_enc, zeta_1_enc, zeta_2_enc, a_plain, l, alice_paillier, out)
        if trace is not None:
            trace.z_enc, trace.z, trace.d_enc, trace.beta_enc = z_enc, b_plain.z, d_enc, beta_enc
            trace.c_step4h, trace.c_sent, trace.delta_b = c_h, c_sent, delta_b
            trace.zeta_1_enc, trace.zeta_2_enc, trace.delta_b_enc = zeta_1_enc, zeta_2_enc, delta_b_enc
        return result
    # five library calls per batch (include/sc_amd.h, scheme-level entry points)
    # Alice: steps 1, 3 (+ the randomization of [[z]])
    z_enc, a_plain = Initiator.step_1_batch(x_enc, y_enc, l, alice_paillier, draws.r, draws.rho_z if randomize else None)
    # Bob: steps 2, 4a, 4b (+ l + 1 randomizations)
    b_plain, d_enc, beta_enc = KeyHolder.step_2_4b_batch(z_enc, l, bob_paillier, bob_dgk, draws.r_bob_dgk if randomize else None)
    # Alice: steps 4c-4i (+ l + 1 randomizations, shuffle)
    c_sent, c_h = Initiator.step_4_batch(d_enc, beta_enc, a_plain, draws.delta_a, alice_dgk, draws.rhos, draws.permutation,
                                         draws.r_alice_dgk if randomize else None, want_unblinded=trace is not None)
    # Bob: steps 4j, 5 (+ 3 randomizations)
    rho3 = cat_rows([draws.rho_zeta_1, draws.rho_zeta_2, draws.rho_delta_b]) if randomize else None   # three blocks of one array: a view
    delta_b, zeta_1_enc, zeta_2_enc, delta_b_enc = KeyHolder.step_4j_5_batch(c_sent, b_plain, bob_paillier, bob_dgk, rho3)
    # Alice: steps 6, 7
    result = Initiator.step_6_7_batch(draws.delta_a, delta_b_enc, zeta_1_enc, zeta_2_enc, a_plain, l, alice_paillier, out)
    if trace is not None:
        trace.z_enc, trace.z, trace.d_enc, trace.beta_enc = z_enc, b_plain.z, d_enc, beta_enc
        trace.c_step4h, trace.c_sent, trace.delta_b = c_h, c_sent, delta_b
        trace.zeta_1_enc, trace.zeta_2_enc, trace.delta_b_enc = zeta_1_enc, zeta_2_enc, delta_b_enc
    return result


def _secure_comparison_batch_pooled(x_enc, y_enc, l, alice_paillier, alice_dgk, bob_paillier, bob_dgk, draws: BatchDraws) -> torch.Tensor:
    count = x_enc.shape[0]
    z_enc, a_plain = Initiator.step_1_batch(x_enc, y_enc, l, alice_paillier, draws.r)
    z_enc = alice_paillier.randomize_from_pool_batch(z_enc)
    b_plain = KeyHolder.step_2_batch(z_enc, l, bob_paillier)
    d_enc, beta_enc = KeyHolder.step_4a_4b_batch(b_plain, l, bob_dgk, bob_paillier, None)
    nw = d_enc.shape[-1]
    rnd = bob_dgk.randomize_from_pool_batch(cat_rows([d_enc.reshape(1, count, nw), beta_enc]).reshape((l + 1) * count, nw))
    rnd = rnd.reshape(l + 1, count, nw)
    d_enc, beta_enc = rnd[0].contiguous(), rnd[1:].contiguous()
    c_h = Initiator.step_4c_to_4h_batch(d_enc, beta_enc, a_plain, draws.delta_a, alice_dgk)
    c = Initiator.step_4i_batch(c_h, alice_dgk, draws.rhos, draws.permutation, None)
    c_sent = alice_dgk.randomize_from_pool_batch(c.reshape((l + 1) * count, nw)).reshape(l + 1, count, nw)
    delta_b = KeyHolder.step_4j_batch(c_sent, bob_dgk)
    triple = bob_paillier.randomize_from_pool_batch(cat_rows(KeyHolder.step_5_batch(b_plain, delta_b, bob_paillier)))
    return Initiator.step_6_7_batch(draws.delta_a, triple[2 * count:], triple[:count], triple[count:2 * count], a_plain, l, alice_paillier)


# ---------------------------------------------------------------------------------------------------------------------------
# Concurrent shards on one GPU.  A batch step has latency-bound stretches (the upper levels of the inversion trees, the
# single-wave extended GCD at their top, the short step 1 / 6 / 7 launches) during which most of the chip idles.  Two (or more)
# shards of the batch, each with its own library context and HIP stream and driven by its own host thread, overlap those
# stretches of one shard with the wide launches of the other: +4.5 % throughput at B = 65536 on one MI355X with two shards.
# ---------------------------------------------------------------------------------------------------------------------------
@dataclass
class PartySet:
    """Both parties' scheme objects bound to one library context (engine) and the stream that context works on."""

    alice_paillier: Paillier
    alice_dgk: DGK
    bob_paillier: Paillier
    bob_dgk: DGK
    stream: "torch.cuda.Stream"
    side: "PartySet | None" = None      # a second context + stream for the randomizer exponentiations (secure_comparison_batch)


def split_draws(draws: BatchDraws, bounds: list[tuple[int, int]]) -> list[BatchDraws]:
    """Contiguous per-shard copies of every random input (per-bit arrays are bit-major, so a shard is not a view); fields that are
    None -- the other party's -- stay None.  The key holder's three randomizer inputs of a shard stay the row blocks of ONE array,
    so that the step joins them without a copy."""
    rows = lambda t, a, b: None if t is None else t[a:b].contiguous()                 # noqa: E731
    planes = lambda t, a, b: None if t is None else t[:, a:b].contiguous()            # noqa: E731
    out = []
    for a, b in bounds:
        n = b - a
        z1 = z2 = db = None
        if draws.rho_zeta_1 is not None:
            rho = torch.cat([draws.rho_zeta_1[a:b], draws.rho_zeta_2[a:b], draws.rho_delta_b[a:b]], dim=0)
            z1, z2, db = rho[:n], rho[n:2 * n], rho[2 * n:]
        out.append(BatchDraws(
            r=rows(draws.r, a, b), delta_a=rows(draws.delta_a, a, b), rhos=planes(draws.rhos, a, b), permutation=rows(draws.permutation, a, b),
            rho_z=rows(draws.rho_z, a, b), r_bob_dgk=planes(draws.r_bob_dgk, a, b), r_alice_dgk=planes(draws.r_alice_dgk, a, b),
            rho_zeta_1=z1, rho_zeta_2=z2, rho_delta_b=db))
    return out


class ConcurrentShards:
    """Runs secure_comparison_batch on several shards at once, one host thread + HIP stream + library context per shard.

    `parties[i]` must have been built on its own Engine (one sc_ctx each: a context orders its work on one stream and reuses its
    temporary buffers from call to call).  Results are handed back on the caller's current stream."""

    def __init__(self, parties: list[PartySet]) -> None:
        from concurrent.futures import ThreadPoolExecutor

        if not parties:
            raise ValueError("at least one party set is required")
        engines = [id(p.alice_paillier.engine) for p in parties]
        if len(set(engines)) != len(engines):
            raise ValueError("every shard needs its own engine (library context)")
        self.parties = parties
        for p in parties:     # every shard's launches share the chip with the other shards' (batch-size policies of the library)
            for scheme in (p.alice_paillier, p.alice_dgk, p.bob_paillier, p.bob_dgk):
                if hasattr(scheme.engine, "set_chip_share"):
                    scheme.engine.set_chip_share(len(parties))
        self._pool = ThreadPoolExecutor(max_workers=len(parties), thread_name_prefix="sc-shard")

    def close(self) -> None:
        self._pool.shutdown(wait=True)
        for p in self.parties:     # the same set of engines __init__ touched
            for scheme in (p.alice_paillier, p.alice_dgk, p.bob_paillier, p.bob_dgk):
                if hasattr(scheme.engine, "set_chip_share"):
                    scheme.engine.set_chip_share(1)

    def run(self, shards: list[tuple[torch.Tensor, torch.Tensor, BatchDraws]], l: int, randomize: bool | str = True,
            out: torch.Tensor | None = None) -> list[torch.Tensor] | torch.Tensor:
        """shards[i] = (x_enc, y_enc, draws) of shard i; returns the per-shard [[x <= y]] arrays -- or, with `out` ([B][2nw], B = the
        shards' sizes together), writes every shard's rows into its block of `out` and returns `out`: the whole batch's result in
        one array without a concatenation pass."""
        if len(shards) != len(self.parties):
            raise ValueError("one shard per party set")
        caller = torch.cuda.current_stream()
        device = caller.device
        blocks = [None] * len(shards)
        if out is not None:
            sizes = [s[0].shape[0] for s in shards]
            if out.dim() != 2 or out.shape[0] != sum(sizes) or not out.is_contiguous():
                raise ValueError(f"out: expected a contiguous [{sum(sizes)}][words] array")
            start = 0
            for i, n in enumerate(sizes):
                blocks[i] = out[start:start + n]
                start += n

        def work(p: PartySet, shard, block):
            with torch.cuda.device(device), torch.cuda.stream(p.stream):
                p.stream.wait_stream(caller)          # the inputs were produced on the caller's stream
                res = secure_comparison_batch(shard[0], shard[1], l, p.alice_paillier, p.alice_dgk, p.bob_paillier, p.bob_dgk,
                                              shard[2], randomize, side=p.side, out=block)
                res.record_stream(caller)             # the caller consumes it on its own stream
                return res

        futures = [self._pool.submit(work, p, s, blk) for p, s, blk in zip(self.parties, shards, blocks)]
        results = [f.result() for f in futures]       # re-raises a shard's exception here
        for p in self.parties:
            caller.wait_stream(p.stream)
        return results if out is None else out
