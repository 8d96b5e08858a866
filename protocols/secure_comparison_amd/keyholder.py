"""Bob -- the party that holds the Paillier and DGK secret keys.

Same constructor, properties, coroutine and static step names as the reference's KeyHolder
(/root/reference/src/tno/mpc/protocols/secure_comparison/keyholder.py, cited as SC/keyholder.py:line), plus
`*_batch` twins on device arrays.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import cast

import torch

from .communicator import Communicator
from .keygen import next_prime
from .schemes import DGK, DGKCiphertext, Paillier, PaillierCiphertext
from .utils import to_bits


@dataclass
class BobPlain:
    """Plaintext-side values Bob derives from z (one entry per comparison)."""

    z: torch.Tensor        # [B][nw]
    beta: torch.Tensor     # [B] u64  z mod 2^l                           (SC/keyholder.py:196)
    d: torch.Tensor        # [B] u64  [z < (N-1)//2]                      (:213)
    zeta_1: torch.Tensor   # [B][nw]  z div 2^l                           (:274)
    zeta_2: torch.Tensor   # [B][nw]  (z + N) div 2^l if d else z div 2^l (:275-282)


class KeyHolder:
    """Player Bob."""

    background_randomness = True     # single comparisons: Paillier randomizers generated beside the critical path (second context)

    fuse_steps = True      # see Initiator.fuse_steps
    coalesce_sessions = True      # see Initiator.coalesce_sessions
    coalesce_max_batch = 4096
    coalesce_linger_s = 0.0
    coalesce_pause_collector_s = 0.5      # see Initiator.coalesce_pause_collector_s

    def __init__(self, l_maximum_bit_length: int, communicator: Communicator | None = None, other_party: str = "",
                 scheme_paillier: Paillier | None = None, scheme_dgk: DGK | None = None, session_id: int = 0) -> None:
        self.l_maximum_bit_length = l_maximum_bit_length
        self.communicator = communicator
        self.other_party = other_party
        self._scheme_paillier = scheme_paillier
        self._scheme_dgk = scheme_dgk
        self.session_id = session_id

    @property
    def scheme_paillier(self) -> Paillier:
        if self._scheme_paillier is None:
            raise ValueError("No Paillier scheme has been initialized or received.")
        return self._scheme_paillier

    @property
    def scheme_dgk(self) -> DGK:
        if self._scheme_dgk is None:
            raise ValueError("No DGK scheme has been initialized or received.")
        return self._scheme_dgk

    # ------------------------------------------------------------------ interactive protocol (one comparison)
    async def perform_secure_comparison(self) -> None:
        """All of Bob's steps with the message exchange of SC/keyholder.py:70-133."""
        if self.communicator is None:
            raise ValueError("Communicator not properly initialized.")
        self.session_id += 1
        sid = self.session_id
        await self.make_and_send_encryption_schemes(sid)
        bind = getattr(self.communicator, "bind_schemes", None)
        if bind is not None:                       # a byte transport builds the ciphertext objects it delivers on these schemes
            bind(self.scheme_paillier, self.scheme_dgk)
        if self.fuse_steps and self.coalesce_sessions:
            with self._coalescer().session():
                await self._perform_coalesced(sid)
            return
        self._start_randomness_generation()
        l = self.l_maximum_bit_length
        pai, dgk = self.scheme_paillier, self.scheme_dgk

        z_enc = await self.communicator.recv(self.other_party, msg_id=f"step_1_session_{sid}")
        if self.fuse_steps:
            await self._perform_fused(z_enc, sid)
            return
        z, beta = KeyHolder.step_2(z_enc, l, pai)
        d_enc = KeyHolder.step_4a(z, dgk, pai, l)
        beta_is_enc = KeyHolder.step_4b(beta, l, dgk)
        d_enc.randomize()
        for b in beta_is_enc:
            b.randomize()
        await self.communicator.send(self.other_party, (d_enc, beta_is_enc), msg_id=f"step_4b_session_{sid}")
        c_is_enc = await self.communicator.recv(self.other_party, msg_id=f"step_4i_session_{sid}")
        delta_b = KeyHolder.step_4j(c_is_enc, dgk)
        zeta_1_enc, zeta_2_enc, delta_b_enc = KeyHolder.step_5(z, l, delta_b, pai)
        for ct in (zeta_1_enc, zeta_2_enc, delta_b_enc):
            ct.randomize()
        await self.communicator.send(self.other_party, (zeta_1_enc, zeta_2_enc, delta_b_enc), msg_id=f"step_5_session_{sid}")

    async def _perform_fused(self, z_enc: PaillierCiphertext, sid: int) -> None:
        """Bob's steps as TWO library calls on one-element batches (sc_keyholder_step2_4b, sc_keyholder_step4j_5) with the
        randomizers taken from the pools in the order the single `.randomize()` calls take them: identical ciphertexts."""
        pai, dgk, l = self.scheme_paillier, self.scheme_dgk, self.l_maximum_bit_length
        e = pai.engine
        nw, nd = pai.mod_n.nwords, dgk.mod_n.nwords
        rand = e.upload([dgk.get_randomness() for _ in range(l + 1)], nd).reshape(l + 1, 1, nd)       # [d] first, then [beta_i] (:106-108)
        plain, d, beta = KeyHolder.step_2_4b_batch(e.upload([z_enc.peek_value()], 2 * nw), l, pai, dgk, rand, randomizers_ready=True)
        d_enc = DGKCiphertext(e.download(d)[0], dgk, fresh=True)
        beta_is_enc = [DGKCiphertext(v, dgk, fresh=True) for v in e.download(beta.reshape(l, nd))]
        await self.communicator.send(self.other_party, (d_enc, beta_is_enc), msg_id=f"step_4b_session_{sid}")
        c_is_enc = await self.communicator.recv(self.other_party, msg_id=f"step_4i_session_{sid}")
        if len(c_is_enc) != l + 1:
            raise ValueError(f"received {len(c_is_enc)} blinded values, expected {l + 1}")
        c = e.upload([ct.peek_value() for ct in c_is_enc], nd).reshape(l + 1, 1, nd)
        rho3 = e.upload([pai.get_randomness() for _ in range(3)], 2 * nw)                                # zeta_1, zeta_2, delta_B (:126-128)
        _, z1, z2, db = KeyHolder.step_4j_5_batch(c, plain, pai, dgk, rho3, randomizers_ready=True)
        out = tuple(PaillierCiphertext(e.download(t)[0], pai, fresh=True) for t in (z1, z2, db))
        await self.communicator.send(self.other_party, out, msg_id=f"step_5_session_{sid}")

    # ------------------------------------------------------------------ concurrent sessions, coalesced into batch launches
    def _coalescer(self):
        from .coalesce import StepCoalescer

        co = self.__dict__.get("_step_coalescer")
        if co is None or (co.max_batch, co.linger_s, co.pause_collector_s) != (self.coalesce_max_batch, self.coalesce_linger_s, self.coalesce_pause_collector_s):
            co = self.__dict__["_step_coalescer"] = StepCoalescer(self.coalesce_max_batch, self.coalesce_linger_s, self.coalesce_pause_collector_s)
        return co

    def _draws(self):
        """See Initiator._draws."""
        src = self.__dict__.get("draw_source")
        if src is None:
            eng = self.scheme_paillier.engine            # one buffered stream per engine: players come and go (one pair per connection),
            src = eng.__dict__.get("_host_draws")        # a refill is a megabyte
            if src is None:
                from .host_draws import HostDraws

                src = eng.__dict__["_host_draws"] = HostDraws.from_engine(eng)
        return src

    async def _perform_coalesced(self, sid: int) -> None:
        """One session of SC/keyholder.py:70-133 whose steps run inside the batch launches it shares with the other sessions in
        flight (Initiator._perform_coalesced).  The session draws the inputs of its 3 + (l + 1) randomizers where
        _start_randomness_generation draws them (:174-179); the k-th `.randomize()` takes what the pool would have handed it."""
        from .coalesce import rows_of

        pai, dgk, l = self.scheme_paillier, self.scheme_dgk, self.l_maximum_bit_length
        n = pai.public_key.n
        co, src = self._coalescer(), self._draws()
        rho = [1 + src.randbelow(n - 1) for _ in range(3)]                          # boot_randomness_generation(3) (:126-128)
        r_dgk = src.bits_rows(dgk.randomizer_bits, l + 1)                           # boot_randomness_generation(l + 1) (:106-108)
        rho3 = rho[::-1]                        # the pool is used from its end: [[zeta_1]], [[zeta_2]], [[delta_B]] take draws 2, 1, 0
        ahead = None
        if self.background_randomness:          # the three Paillier randomizers are not needed before step 5: queued on the second context now
            ahead = await co.submit("randomizers_step_5", self._run_randomizers_ahead, rho3, first=True)
        z_enc = await self.communicator.recv(self.other_party, msg_id=f"step_1_session_{sid}")
        z_row = rows_of([z_enc], 2 * pai.mod_n.nwords)[0]
        d_enc, beta_is_enc, zetas = await co.submit("step_2_4b", self._run_step_2_4b, (z_row, r_dgk[::-1]), first=ahead is None)   # [d] first, then [beta_i]
        await self.communicator.send(self.other_party, (d_enc, beta_is_enc), msg_id=f"step_4b_session_{sid}")
        c_is_enc = await self.communicator.recv(self.other_party, msg_id=f"step_4i_session_{sid}")
        if len(c_is_enc) != l + 1:
            raise ValueError(f"received {len(c_is_enc)} blinded values, expected {l + 1}")
        three = await co.submit("step_4j_5", self._run_step_4j_5, (rows_of(c_is_enc, dgk.mod_n.nwords), zetas, rho3, ahead))
        await self.communicator.send(self.other_party, three, msg_id=f"step_5_session_{sid}")

    def _run_randomizers_ahead(self, items: list) -> list:
        """rho^N mod N^2 for the 3 K randomizer bases of K sessions, QUEUED on the scheme's background context (the reference starts
        background workers for them, SC/keyholder.py:174-179) -- returns at once; step 5 collects.  Where there is no second context
        (the CPU tier's stand-in engine) the sessions get None and step 5 computes the randomizers itself."""
        job = self.scheme_paillier._launch_randomness_values([v for it in items for v in it])
        if job is None:
            return [None] * len(items)
        shared = {"job": job, "rows": None}
        return [(shared, 3 * i) for i in range(len(items))]

    def _run_step_2_4b(self, items: list) -> list:
        """Steps 2, 4a, 4b + the l + 1 `.randomize()` of K sessions: one sc_keyholder_step2_4b call; per session ([d], [[beta_i]] as
        fresh ciphertexts, and the rows of the two plaintext quotients zeta_1, zeta_2, which travel to step 5 with the session)."""
        import numpy as np

        from .limbs import RowBlock

        pai, dgk, l = self.scheme_paillier, self.scheme_dgk, self.l_maximum_bit_length
        e, nw, k = pai.engine, pai.mod_n.nwords, len(items)
        er = (dgk.randomizer_bits + 31) // 32
        z, exps = np.empty((k, 2 * nw), dtype="<u4"), np.empty((l + 1, k, er), dtype="<u4")
        for b, it in enumerate(items):
            z[b], exps[:, b] = it[0], it[1]
        assert dgk.public_key.u > (1 << (l + 2))
        # (KeyHolder.step_2_4b_batch, keeping [d] and the planes [beta_i] as the ONE array [l+1][K][nd] the launch wrote: a session's
        # l + 1 ciphertexts are then the rows of one block, which the initiator's next batched call takes back as it is)
        _, _, _, zeta_1, zeta_2, enc = e.keyholder_step2_4b(pai.key, dgk.key, l, e.upload_words(z), e.upload_words(exps).reshape((l + 1) * k, er), False, None)
        planes = e.download_words(enc)
        z1, z2 = e.download_words(zeta_1), e.download_words(zeta_2)
        pub = dgk.for_wire()
        out = []
        for b in range(k):
            cts = DGKCiphertext.rows(RowBlock(planes, b), pub, fresh=True)
            out.append((cts[0], cts[1:], (z1[b], z2[b])))
        return out

    def _run_step_4j_5(self, items: list) -> list:
        """Steps 4j, 5 + the three `.randomize()` of K sessions: one sc_keyholder_step4j_5 call (the randomizers finished ahead of
        time where every session of the batch has them, else computed in the call); per session the three fresh ciphertexts."""
        import numpy as np

        from .coalesce import int_rows, stack_blocks
        from .limbs import RowBlock

        pai, dgk, l = self.scheme_paillier, self.scheme_dgk, self.l_maximum_bit_length
        e, nw, nd, k = pai.engine, pai.mod_n.nwords, dgk.mod_n.nwords, len(items)
        c, zeta = stack_blocks([it[0] for it in items], l + 1, nd), np.empty((2, k, nw), dtype="<u4")      # the initiator's own array when the sessions are the same
        for b, it in enumerate(items):
            zeta[0, b], zeta[1, b] = it[1][0], it[1][1]
        tz = e.upload_words(zeta)
        ready = all(it[3] is not None for it in items)
        if ready:
            rho = np.empty((3, k, 2 * nw), dtype="<u4")
            for b, it in enumerate(items):
                shared, at = it[3]
                if shared["rows"] is None:                # first session of that background batch to get here: wait for it, read it once
                    t, ev, twin = shared["job"]
                    ev.synchronize()
                    shared["rows"] = twin.engine.download_words(t)
                rho[:, b] = shared["rows"][at:at + 3]
            rho3 = e.upload_words(rho).reshape(3 * k, 2 * nw)
        else:
            rho3 = e.upload_words(int_rows([it[2][j] for j in range(3) for it in items], nw))
        _, z1, z2, db = KeyHolder.step_4j_5_batch(e.upload_words(c), BobPlain(None, None, None, tz[0], tz[1]), pai, dgk, rho3, randomizers_ready=ready)
        out3 = np.stack([e.download_words(z1), e.download_words(z2), e.download_words(db)])       # [3][K][2nw]
        pub = pai.for_wire()
        return [tuple(PaillierCiphertext.rows(RowBlock(out3, b), pub, fresh=True)) for b in range(k)]

    async def perform_secure_comparison_batch(self, draws=None, source: str = "device", generator=None) -> None:
        """Bob's side of Initiator.perform_secure_comparison_batch.  `draws` (batch.BatchDraws; Bob's fields) injects
        the randomizer inputs; otherwise the 3 Paillier + (l+1) DGK randomizer inputs per comparison (SC/keyholder.py:174-179
        scaled by B) are drawn on the device by the engine's CSPRNG and consumed by the same fused launches.  When Alice
        announces a chunked batch (wire.pack_plan as the first message) the chunks are served as concurrent sub-sessions."""
        import asyncio

        from . import wire
        from .batch import split_draws

        if self.communicator is None:
            raise ValueError("Communicator not properly initialized.")
        comm = self.communicator
        self.session_id += 1
        sid = self.session_id
        pai, dgk = self.scheme_paillier, self.scheme_dgk
        await comm.send(self.other_party, wire.pack_public_schemes(pai, dgk), msg_id=f"schemes_batch_session_{sid}")
        first = await comm.recv(self.other_party, msg_id=f"step_1_batch_session_{sid}")
        sizes = wire.plan_of(first)
        if sizes is None:
            await self._batch_session(f"session_{sid}", first, draws, source, generator)
            return
        parts = [None] * len(sizes)
        if draws is not None:
            if draws.r_bob_dgk.shape[1] != sum(sizes):
                raise ValueError(f"draws for {draws.r_bob_dgk.shape[1]} comparisons, the plan announces {sum(sizes)}")
            bounds, start = [], 0
            for n in sizes:
                bounds.append((start, start + n))
                start += n
            parts = split_draws(draws, bounds)
        await asyncio.gather(*(self._batch_session(f"session_{sid}_chunk_{i}", None, parts[i], source, generator, expect_count=n)
                               for i, n in enumerate(sizes)))

    async def _batch_session(self, tag: str, first, draws, source: str, generator, expect_count: int | None = None) -> None:
        """One (sub-)session: Bob's steps around the four message exchanges with message ids `.._{tag}`; `first` is the step-1
        message when it has been received already."""
        from . import wire
        from ._views import cat_rows
        from .batch import draw_bob

        comm, pai, dgk, l = self.communicator, self.scheme_paillier, self.scheme_dgk, self.l_maximum_bit_length
        dev = pai.engine.device
        if first is None:
            first = await comm.recv(self.other_party, msg_id=f"step_1_batch_{tag}")
        (z_enc,) = wire.incoming(first, dev, expect=1)
        if not isinstance(z_enc, torch.Tensor) or z_enc.dim() != 2:
            raise ValueError(f"[[z]]: received shape {tuple(getattr(z_enc, 'shape', ()))}, expected [B][{2 * pai.mod_n.nwords}]")
        count = z_enc.shape[0]                       # the batch size is Alice's to choose; everything else is checked against it
        if expect_count is not None and count != expect_count:
            raise ValueError(f"[[z]]: {count} comparisons in a chunk the plan announced with {expect_count}")
        z_enc = wire.expect_array(z_enc, (count, 2 * pai.mod_n.nwords), "[[z]]")
        if draws is None:
            draws = draw_bob(count, l, pai, dgk, source, generator)
        # (byte transports: the messages are written in place by the steps' last launches, see Initiator._batch_session)
        msg = wire.reserve(comm, dev, (l + 1, count, dgk.mod_n.nwords))
        plain, d_enc, beta_enc = KeyHolder.step_2_4b_batch(z_enc, l, pai, dgk, draws.r_bob_dgk, out=None if msg is None else msg.arrays[0])
        await comm.send(self.other_party, wire.outgoing(comm, d_enc, beta_enc) if msg is None else await msg.finish(), msg_id=f"step_4b_batch_{tag}")
        (c_enc,) = wire.incoming(await comm.recv(self.other_party, msg_id=f"step_4i_batch_{tag}"), dev, expect=1)
        c_enc = wire.expect_array(c_enc, (l + 1, count, dgk.mod_n.nwords), "[c_i]")
        msg = wire.reserve(comm, dev, (3, count, 2 * pai.mod_n.nwords))
        _, zeta_1_enc, zeta_2_enc, delta_b_enc = KeyHolder.step_4j_5_batch(
            c_enc, plain, pai, dgk, cat_rows([draws.rho_zeta_1, draws.rho_zeta_2, draws.rho_delta_b]),
            out=None if msg is None else msg.arrays[0].reshape(3 * count, 2 * pai.mod_n.nwords))
        await comm.send(self.other_party, wire.outgoing(comm, zeta_1_enc, zeta_2_enc, delta_b_enc) if msg is None else await msg.finish(),
                        msg_id=f"step_5_batch_{tag}")

    async def make_and_send_encryption_schemes(self, session_id: int = 1, key_length_paillier: int = 2048,
                                               v_bits_dgk: int = 160, n_bits_dgk: int = 2048) -> None:
        """Create missing schemes (defaults of SC/keyholder.py:138-140, u = next_prime(2^(l+2)) :164) and send the
        public parts to Alice (:168-172)."""
        if self.communicator is None:
            raise ValueError("Communicator not properly initialized.")
        if self._scheme_paillier is None:
            self._scheme_paillier = Paillier.from_security_parameter(key_length=key_length_paillier)
        if self._scheme_dgk is None:
            self._scheme_dgk = DGK.from_security_parameter(v_bits=v_bits_dgk, n_bits=n_bits_dgk,
                                                           u=next_prime(1 << (self.l_maximum_bit_length + 2)),
                                                           full_decryption=False)
        await self.communicator.send(self.other_party, (self.scheme_paillier, self.scheme_dgk),
                                     msg_id=f"schemes_session_{session_id}")

    def _start_randomness_generation(self) -> None:
        """3 Paillier + (l+1) DGK randomizers (SC/keyholder.py:174-179)."""
        # the three Paillier randomizers are not needed before step 5: generated beside the initiator's first message and
        # Bob's own steps 2 .. 4b on a second context (the reference starts background workers here); the DGK ones are needed at
        # once and cost a fixed-base launch
        self.scheme_paillier.boot_randomness_generation(3, background=self.background_randomness)
        self.scheme_dgk.boot_randomness_generation(self.l_maximum_bit_length + 1)

    # ------------------------------------------------------------------ single-ciphertext steps
    @staticmethod
    def step_2(z_enc: PaillierCiphertext, l: int, scheme_paillier: Paillier) -> tuple[int, int]:
        """z = Dec([[z]]) without decoding, beta = z mod 2^l (SC/keyholder.py:181-196)."""
        z = cast(int, scheme_paillier.decrypt(z_enc, apply_encoding=False))
        return z, z % (1 << l)

    @staticmethod
    def step_4a(z: int, scheme_dgk: DGK, scheme_paillier: Paillier, l: int) -> DGKCiphertext:
        """[d], d = (z < (N-1)/2) (SC/keyholder.py:198-216)."""
        assert scheme_dgk.public_key.u > (1 << (l + 2))
        return scheme_dgk.unsafe_encrypt(int(z < (scheme_paillier.public_key.n - 1) // 2), apply_encoding=False)

    @staticmethod
    def step_4b(beta: int, l: int, scheme_dgk: DGK) -> list[DGKCiphertext]:
        """[beta_i], i = 0..l-1 (SC/keyholder.py:218-233)."""
        return [scheme_dgk.unsafe_encrypt(b, apply_encoding=False) for b in to_bits(beta, l)]

    @staticmethod
    def step_4j(c_is_enc: list[DGKCiphertext], scheme_dgk: DGK) -> int:
        """delta_B = 1 iff some c_i decrypts to zero (SC/keyholder.py:235-253); one batched zero test."""
        e = scheme_dgk.engine
        flags = scheme_dgk.is_zero_batch(e.upload([c.peek_value() for c in c_is_enc], scheme_dgk.mod_n.nwords))
        return int(bool(flags.any().item()))

    @staticmethod
    def step_5(z: int, l: int, delta_b: int, scheme_paillier: Paillier) -> tuple[PaillierCiphertext, PaillierCiphertext, PaillierCiphertext]:
        """[[zeta_1]], [[zeta_2]], [[delta_B]] (SC/keyholder.py:255-287)."""
        n = scheme_paillier.public_key.n
        enc = lambda m: scheme_paillier.unsafe_encrypt(m, apply_encoding=False)  # noqa: E731
        zeta_2 = (z + n) >> l if z < (n - 1) // 2 else z >> l
        return enc(z >> l), enc(zeta_2), enc(delta_b)

    # ------------------------------------------------------------------ batched steps
    @staticmethod
    def step_2_batch(z_enc: torch.Tensor, l: int, scheme_paillier: Paillier) -> BobPlain:
        """Decrypt B ciphertexts and derive beta, d, zeta_1, zeta_2."""
        z = scheme_paillier.decrypt_raw_batch(z_enc)
        beta, d, zeta_1, zeta_2 = scheme_paillier.engine.plain_bob(z, scheme_paillier.public_key.n, l)
        return BobPlain(z, beta, d, zeta_1, zeta_2)

    @staticmethod
    def step_4a_4b_batch(plain: BobPlain, l: int, scheme_dgk: DGK, scheme_paillier: Paillier,
                         randomizer_exponents: torch.Tensor | None = None) -> tuple[torch.Tensor, torch.Tensor]:
        """[d] ([B][nw]) and [beta_i] ([l][B][nw], bit-major).  With `randomizer_exponents` ([l+1][B][ew], row 0 for d,
        row 1+i for beta_i) the `.randomize()` calls of SC/keyholder.py:106-108 are fused in: g^bit * h^r."""
        assert scheme_dgk.public_key.u > (1 << (l + 2))
        count = plain.beta.shape[0]
        shifts = torch.arange(l, device=plain.beta.device, dtype=torch.int64).reshape(l, 1)
        bits = torch.cat([plain.d.reshape(1, count), (plain.beta.reshape(1, count) >> shifts) & 1], dim=0)  # [l+1][B]
        if randomizer_exponents is not None:
            enc = scheme_dgk.encrypt_bits_randomized_batch(bits.reshape(-1), randomizer_exponents.reshape((l + 1) * count, -1))
        else:
            enc = scheme_dgk.encrypt_bits_batch(bits.reshape(-1))
        enc = enc.reshape(l + 1, count, -1)
        return enc[0].contiguous(), enc[1:].contiguous()

    @staticmethod
    def step_2_4b_batch(z_enc: torch.Tensor, l: int, scheme_paillier: Paillier, scheme_dgk: DGK,
                        randomizer_exponents: torch.Tensor | None = None, randomizers_ready: bool = False,
                        out: torch.Tensor | None = None) -> tuple[BobPlain, torch.Tensor, torch.Tensor]:
        """Steps 2, 4a, 4b (and, with `randomizer_exponents` [l+1][B][ew], the l + 1 `.randomize()` of SC/keyholder.py:106-108) in ONE
        library call (sc_keyholder_step2_4b).  Returns (plain, [d] as [B][nw], [beta_i] as [l][B][nw]) -- the latter two are the
        planes of one array, so the initiator's inversion pass takes them without a copy."""
        assert scheme_dgk.public_key.u > (1 << (l + 2))
        count = z_enc.shape[0]
        rr = None if randomizer_exponents is None else randomizer_exponents.reshape((l + 1) * count, -1)
        z, beta, d, zeta_1, zeta_2, enc = scheme_paillier.engine.keyholder_step2_4b(scheme_paillier.key, scheme_dgk.key, l, z_enc, rr,
                                                                                    randomizers_ready, out)
        return BobPlain(z, beta, d, zeta_1, zeta_2), enc[0], enc[1:]

    @staticmethod
    def step_4j_5_batch(c_is_enc: torch.Tensor, plain: BobPlain, scheme_paillier: Paillier, scheme_dgk: DGK,
                        rho3: torch.Tensor | None = None, randomizers_ready: bool = False,
                        out: torch.Tensor | None = None) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """Steps 4j and 5 (and, with rho3 [3B][nw] = the bases for [[zeta_1]], [[zeta_2]], [[delta_B]] in that order, the three
        `.randomize()` of SC/keyholder.py:126-128) in ONE library call (sc_keyholder_step4j_5).
        Returns (delta_B [B] u64, [[zeta_1]], [[zeta_2]], [[delta_B]])."""
        l, count = c_is_enc.shape[0] - 1, c_is_enc.shape[1]
        delta_b, enc = scheme_paillier.engine.keyholder_step4j_5(scheme_paillier.key, scheme_dgk.key, l, c_is_enc, plain.zeta_1, plain.zeta_2, rho3,
                                                                 randomizers_ready, out)
        return delta_b, enc[:count], enc[count:2 * count], enc[2 * count:]

    @staticmethod
    def step_4j_batch(c_is_enc: torch.Tensor, scheme_dgk: DGK) -> torch.Tensor:
        """delta_B per comparison: OR over the bit axis of the zero tests.  c_is_enc: [l+1][B][nw] -> [B] u64."""
        return scheme_dgk.any_zero_batch(c_is_enc)     # the OR over the bit axis happens in the zero-test launch

    @staticmethod
    def step_5_batch(plain: BobPlain, delta_b: torch.Tensor, scheme_paillier: Paillier) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """[[zeta_1]], [[zeta_2]], [[delta_B]] for B comparisons (unrandomized)."""
        nw = plain.zeta_1.shape[-1]
        db_words = torch.zeros((delta_b.shape[0], nw), dtype=torch.int32, device=delta_b.device)
        db_words[:, 0] = delta_b.to(torch.int32)
        enc = scheme_paillier.encrypt_raw_batch(torch.cat([plain.zeta_1, plain.zeta_2, db_words], dim=0))
        count = delta_b.shape[0]
        return enc[:count], enc[count:2 * count], enc[2 * count:]
