"""Alice -- the party that holds [[x]], [[y]] and learns [[x <= y]].

Same constructor, properties, coroutine and static step names as the reference's Initiator
(/root/reference/src/tno/mpc/protocols/secure_comparison/initiator.py, cited as SC/initiator.py:line), with two
differences that make the arithmetic reproducible and batchable:

* every random draw can be injected (`r=`, `delta_a=`, `rhos=`, `permutation=`); when omitted it is drawn from
  `secrets` exactly where the reference draws it;
* each step has a `*_batch` twin working on device arrays of B comparisons (bit-major for the per-bit vectors),
  which is what the GPU path is built for.  The single-ciphertext steps run on the GPU too (batches of one).
"""
from __future__ import annotations

import asyncio
import secrets
from dataclasses import dataclass
from typing import Any, Sequence, cast

import torch

from .communicator import Communicator
from .schemes import DGK, DGKCiphertext, Paillier, PaillierCiphertext
from .utils import to_bits


@dataclass
class AlicePlain:
    """Plaintext-side values Alice derives from her blinding value r (one entry per comparison)."""

    r: torch.Tensor            # [B][nw]   r, 0 <= r < N                       (SC/initiator.py:250)
    alpha: torch.Tensor        # [B] u64   r mod 2^l                           (:270)
    alpha_tilde: torch.Tensor  # [B] u64   (r - N) mod 2^l                     (:373)
    r_small: torch.Tensor      # [B] u64   [r < (N-1)//2]                      (:289, :559)
    r_shift: torch.Tensor      # [B][nw]   r div 2^l                           (:562)


class Initiator:
    """Player Alice."""

    # perform_secure_comparison runs Alice's steps as three library calls on one-element batches (True) or, like the reference's
    # body, step by step through the ciphertext operator algebra (False: one launch per operator).  Same ciphertexts either way.
    fuse_steps = True
    # Concurrent perform_secure_comparison sessions of one Initiator hand their steps to a coalescer (coalesce.StepCoalescer): requests
    # of the same step that arrive in the same turn of the event loop run as ONE call of the batch entry points.  A lone session
    # takes the same path as a batch of one; False restores one library call per session and step.
    coalesce_sessions = True
    coalesce_max_batch = 4096
    coalesce_linger_s = 0.0
    # While sessions are in flight the cyclic garbage collector is paused, for at most this many seconds from the first session of
    # a burst (coalesce._CollectorPause: it finds nothing to collect among a burst's ~10^5 live message objects and costs a third to
    # half of the burst); 0 = the library never touches the collector.
    coalesce_pause_collector_s = 0.5

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
    async def perform_secure_comparison(self, x: PaillierCiphertext | float, y: PaillierCiphertext | float) -> PaillierCiphertext:
        """All of Alice's steps with the message exchange of SC/initiator.py:69-175."""
        if self.communicator is None:
            raise ValueError("Communicator not properly initialized.")
        self.session_id += 1
        sid = self.session_id
        await self.receive_encryption_schemes(sid)
        bind = getattr(self.communicator, "bind_schemes", None)
        if bind is not None:                       # a byte transport builds the ciphertext objects it delivers on these schemes
            bind(self.scheme_paillier, self.scheme_dgk)
        if self.fuse_steps and self.coalesce_sessions:
            with self._coalescer().session():
                return await self._perform_coalesced(x, y, sid)
        self._start_randomness_generation()
        l = self.l_maximum_bit_length
        pai, dgk = self.scheme_paillier, self.scheme_dgk
        x_enc = x if isinstance(x, PaillierCiphertext) else pai.unsafe_encrypt(x)
        y_enc = y if isinstance(y, PaillierCiphertext) else pai.unsafe_encrypt(y)
        if self.fuse_steps:
            return await self._perform_fused(x_enc, y_enc, sid)

        z_enc, r = Initiator.step_1(x_enc, y_enc, l, pai)
        z_enc.randomize()
        await self.communicator.send(self.other_party, z_enc, msg_id=f"step_1_session_{sid}")
        alpha = Initiator.step_3(r, l)
        d_enc, beta_is_enc = await self.communicator.recv(self.other_party, msg_id=f"step_4b_session_{sid}")
        d_enc = Initiator.step_4c(d_enc, r, dgk, pai)
        xor_is_enc = Initiator.step_4d(alpha, beta_is_enc)
        w_is_enc, alpha_tilde = Initiator.step_4e(r, alpha, xor_is_enc, d_enc, pai)
        w_is_enc = Initiator.step_4f(w_is_enc)
        s, delta_a = Initiator.step_4g()
        c_is_enc = Initiator.step_4h(s, alpha, alpha_tilde, d_enc, beta_is_enc, w_is_enc, delta_a, dgk)
        c_is_enc = Initiator.step_4i(c_is_enc, dgk, do_shuffle=True)
        for c in c_is_enc:
            c.randomize()
        await self.communicator.send(self.other_party, c_is_enc, msg_id=f"step_4i_session_{sid}")
        zeta_1_enc, zeta_2_enc, delta_b_enc = await self.communicator.recv(self.other_party, msg_id=f"step_5_session_{sid}")
        beta_lt_alpha_enc = Initiator.step_6(delta_a, delta_b_enc)
        return Initiator.step_7(zeta_1_enc, zeta_2_enc, r, l, beta_lt_alpha_enc, pai)

    async def _perform_fused(self, x_enc: PaillierCiphertext, y_enc: PaillierCiphertext, sid: int) -> PaillierCiphertext:
        """The same exchange with Alice's steps as THREE library calls on one-element batches (sc_initiator_step1 / _step4 /
        _step67) instead of one launch per ciphertext operator of steps 4d-4h (~10 l of them, each with its own upload,
        synchronisation and download): identical ciphertexts -- the random values are drawn in the order the single steps draw
        them (r; delta_A; rho_i; the shuffle), the randomizers come from the same pools, and output k of the shuffle takes the
        randomizer the k-th `.randomize()` of SC/initiator.py:153-154 would have taken."""
        pai, dgk, l = self.scheme_paillier, self.scheme_dgk, self.l_maximum_bit_length
        e = pai.engine
        nw, nd = pai.mod_n.nwords, dgk.mod_n.nwords
        n, u = pai.public_key.n, dgk.public_key.u
        tx, ty = e.upload([x_enc.get_value()], 2 * nw), e.upload([y_enc.get_value()], 2 * nw)
        assert (1 << (l + 2)) < n // 2
        r = secrets.randbelow(n)                                                   # step 1 (SC/initiator.py:250)
        rz = e.upload([pai.get_randomness()], 2 * nw)                               # z_enc.randomize() (:109)
        z, plain = Initiator.step_1_batch(tx, ty, l, pai, e.upload([r], nw), rz, randomizers_ready=True)
        z_enc = PaillierCiphertext(e.download(z)[0], pai, fresh=True)
        await self.communicator.send(self.other_party, z_enc, msg_id=f"step_1_session_{sid}")
        d_enc, beta_is_enc = await self.communicator.recv(self.other_party, msg_id=f"step_4b_session_{sid}")
        if len(beta_is_enc) != l:
            raise ValueError(f"received {len(beta_is_enc)} encrypted bits, expected {l}")
        planes = e.upload([d_enc.get_value()] + [b.get_value() for b in beta_is_enc], nd).reshape(l + 1, 1, nd)
        assert 0 <= r < n                                                          # step 4c (:286-288)
        _, delta_a = Initiator.step_4g()                                           # (:420)
        rhos = [secrets.randbelow(u - 1) + 1 for _ in range(l + 1)]                 # step 4i (:512)
        perm = Initiator.shuffle(list(range(l + 1)))                                # (:516): output k takes the blinded c at perm[k]
        rand = [dgk.get_randomness() for _ in range(l + 1)]                         # the k-th c.randomize() (:153-154) ...
        by_item = [0] * (l + 1)
        for k, src in enumerate(perm):
            by_item[src] = rand[k]                                                  # ... applied to the item that lands at output k
        ew = (u.bit_length() + 31) // 32
        c, _ = Initiator.step_4_batch(planes[0], planes[1:], plain, e.upload_u64([delta_a]), dgk, e.upload(rhos, ew).reshape(l + 1, 1, ew),
                                      e.upload_u64(perm).reshape(1, l + 1), e.upload(by_item, nd).reshape(l + 1, 1, nd), randomizers_ready=True)
        c_is_enc = [DGKCiphertext(v, dgk, fresh=True) for v in e.download(c.reshape(l + 1, nd))]
        await self.communicator.send(self.other_party, c_is_enc, msg_id=f"step_4i_session_{sid}")
        zeta_1_enc, zeta_2_enc, delta_b_enc = await self.communicator.recv(self.other_party, msg_id=f"step_5_session_{sid}")
        three = e.upload([zeta_1_enc.get_value(), zeta_2_enc.get_value(), delta_b_enc.get_value()], 2 * nw)
        res = Initiator.step_6_7_batch(e.upload_u64([delta_a]), three[2:3], three[0:1], three[1:2], plain, l, pai)
        return PaillierCiphertext(e.download(res)[0], pai)

    # ------------------------------------------------------------------ concurrent sessions, coalesced into batch launches
    def _coalescer(self):
        from .coalesce import StepCoalescer

        co = self.__dict__.get("_step_coalescer")
        if co is None or (co.max_batch, co.linger_s, co.pause_collector_s) != (self.coalesce_max_batch, self.coalesce_linger_s, self.coalesce_pause_collector_s):
            co = self.__dict__["_step_coalescer"] = StepCoalescer(self.coalesce_max_batch, self.coalesce_linger_s, self.coalesce_pause_collector_s)
        return co

    def _draws(self):
        """Where a coalesced session takes its random draws from (host_draws.py): `draw_source` when one was injected, else a buffered
        stream of the engine's device generator (the OS generator where the engine has none)."""
        src = self.__dict__.get("draw_source")
        if src is None:
            eng = self.scheme_paillier.engine            # one buffered stream per engine: players come and go (one pair per connection),
            src = eng.__dict__.get("_host_draws")        # a refill is a megabyte
            if src is None:
                from .host_draws import HostDraws

                src = eng.__dict__["_host_draws"] = HostDraws.from_engine(eng)
        return src

    async def _perform_coalesced(self, x: PaillierCiphertext | float, y: PaillierCiphertext | float, sid: int) -> PaillierCiphertext:
        """One session of SC/initiator.py:69-175 whose steps run inside the batch launches it shares with the other sessions in
        flight.  The session draws what the single path draws, where it draws it: the inputs of its 1 + (l + 1) randomizers
        (_start_randomness_generation, :205-210), r (:250), delta_A (:420), rho_i (:512), the shuffle (:223); the k-th
        `.randomize()` takes the randomizer the pool would have handed it (the pool is used from its end).  Plaintext inputs are
        encrypted inside the step-1 launch (the unrandomized encryption 1 + mN of `unsafe_encrypt`, :93-102).  Ciphertext values
        travel as rows of words (coalesce.rows_of): no Python integer is made of them on the way."""
        import numpy as np

        from .coalesce import rows_of

        pai, dgk, l = self.scheme_paillier, self.scheme_dgk, self.l_maximum_bit_length
        n, u = pai.public_key.n, dgk.public_key.u
        co, src = self._coalescer(), self._draws()
        rho_z = 1 + src.randbelow(n - 1)                                           # boot_randomness_generation(1): [[z]].randomize() (:109)
        r_dgk = src.bits_rows(dgk.randomizer_bits, l + 1)                          # boot_randomness_generation(l + 1) (:153-154)
        # (ciphertext, None) or (None, encoded plaintext): plaintext inputs are encrypted inside the batched call
        xv = (x.consume(), None) if isinstance(x, PaillierCiphertext) else (None, pai._encode(x))
        yv = (y.consume(), None) if isinstance(y, PaillierCiphertext) else (None, pai._encode(y))
        assert (1 << (l + 2)) < n // 2
        r = src.randbelow(n)                                                       # step 1 (:250)
        z_enc, plain = await co.submit("step_1", self._run_step_1, (xv, yv, r, rho_z), first=True)
        await self.communicator.send(self.other_party, z_enc, msg_id=f"step_1_session_{sid}")
        d_enc, beta_is_enc = await self.communicator.recv(self.other_party, msg_id=f"step_4b_session_{sid}")
        if len(beta_is_enc) != l:
            raise ValueError(f"received {len(beta_is_enc)} encrypted bits, expected {l}")
        assert 0 <= r < n                                                          # step 4c (:286-288)
        delta_a = src.coin()                                                       # step 4g (:420)
        rhos = src.below_rows_nonzero(u, l + 1)                                    # step 4i (:512): rho_i = 1 + randbelow(u - 1)
        perm = src.permutation(l + 1)                                              # (:516): output k takes the blinded c at perm[k]
        exps = np.empty_like(r_dgk)
        exps[perm] = r_dgk[::-1]              # the k-th c.randomize() (:153-154) pops the pool's last entry: draw l - k, applied to the item that lands at output k
        planes = rows_of([d_enc.consume()] + [b.consume() for b in beta_is_enc], dgk.mod_n.nwords)
        c_is_enc = await co.submit("step_4", self._run_step_4, (planes, plain, delta_a, rhos, perm, exps))
        await self.communicator.send(self.other_party, c_is_enc, msg_id=f"step_4i_session_{sid}")
        three = await self.communicator.recv(self.other_party, msg_id=f"step_5_session_{sid}")
        if len(three) != 3:
            raise ValueError(f"received {len(three)} ciphertexts, expected [[zeta_1]], [[zeta_2]], [[delta_B]]")
        return await co.submit("step_6_7", self._run_step_6_7, (rows_of([t.consume() for t in three], 2 * pai.mod_n.nwords), plain, delta_a))

    def _run_step_1(self, items: list) -> list:
        """Step 1 + `.randomize()` of K sessions: one sc_initiator_step1 call (the pair exponentiation rho_z^N fused in).  Per session:
        ([[z]] as a fresh ciphertext, Alice's plaintext-side values of SC/initiator.py:270, :289, :373, :559-562 -- alpha, alpha~,
        [r < (N-1)/2] and the row of r div 2^l -- for the session's later steps, whose batches need not be this one)."""
        import numpy as np

        from .coalesce import int_rows, rows_of

        pai, l = self.scheme_paillier, self.l_maximum_bit_length
        e, nw, k = pai.engine, pai.mod_n.nwords, len(items)
        xy = np.zeros((2 * k, 2 * nw), dtype="<u4")                                 # rows 0 .. K-1: [[x]], K .. 2K-1: [[y]]
        plain = []
        for c in (0, 1):
            for b, it in enumerate(items):
                ct, m = it[c]
                if ct is None:
                    plain.append((c * k + b, m))
                else:
                    xy[c * k + b] = rows_of([ct], 2 * nw)[0]
        t = e.upload_words(xy)
        if plain:                                     # unsafe_encrypt of the plaintext inputs (SC/initiator.py:93-102), one launch
            rows = torch.tensor([i for i, _ in plain], dtype=torch.int64, device=t.device)
            t[rows] = pai.encrypt_raw_batch(e.upload_words(int_rows([m for _, m in plain], nw)))
        z, p = Initiator.step_1_batch(t[:k], t[k:], l, pai, e.upload_words(int_rows([it[2] for it in items], nw)),
                                      e.upload_words(int_rows([it[3] for it in items], nw)))
        zs, shift = e.download_words(z), e.download_words(p.r_shift)
        alpha, tilde, small = p.alpha.tolist(), p.alpha_tilde.tolist(), p.r_small.tolist()
        zc = PaillierCiphertext.rows(zs, pai.for_wire(), fresh=True)
        return [(zc[b], (alpha[b], tilde[b], small[b], shift[b])) for b in range(k)]

    def _run_step_4(self, items: list) -> list:
        """Steps 4c .. 4i + the l + 1 `.randomize()` of K sessions: one sc_initiator_step4 call on bit-major planes [l+1][K]; per session
        the l + 1 fresh ciphertexts [c_i] as sent."""
        import numpy as np

        from .coalesce import stack_blocks
        from .limbs import RowBlock

        dgk, l = self.scheme_dgk, self.l_maximum_bit_length
        e, nd, k = dgk.engine, dgk.mod_n.nwords, len(items)
        ew, er = (dgk.public_key.u.bit_length() + 31) // 32, (dgk.randomizer_bits + 31) // 32
        planes = stack_blocks([it[0] for it in items], l + 1, nd)            # the key holder's own array when the sessions are the same
        rhos, exps = (np.empty((l + 1, k, w), dtype="<u4") for w in (ew, er))
        for b, it in enumerate(items):
            rhos[:, b], exps[:, b] = it[3], it[5]
        flags = lambda col: e.upload_u64([it[1][col] & 0xFFFFFFFFFFFFFFFF for it in items])   # noqa: E731
        plain = AlicePlain(None, flags(0), flags(1), flags(2), None)
        tp = e.upload_words(planes)
        c, _ = Initiator.step_4_batch(tp[0], tp[1:], plain, e.upload_u64([it[2] for it in items]), dgk, e.upload_words(rhos),
                                      torch.from_numpy(np.array([it[4] for it in items], dtype=np.int64)).to(tp.device), e.upload_words(exps))
        cv, pub = e.download_words(c), dgk.for_wire()
        return [DGKCiphertext.rows(RowBlock(cv, b), pub, fresh=True) for b in range(k)]

    def _run_step_6_7(self, items: list) -> list:
        """Steps 6 and 7 of K sessions: one sc_initiator_step67 call; per session [[x <= y]]."""
        import numpy as np

        from .coalesce import stack_blocks

        pai, l = self.scheme_paillier, self.l_maximum_bit_length
        e, nw, k = pai.engine, pai.mod_n.nwords, len(items)
        three, shift = stack_blocks([it[0] for it in items], 3, 2 * nw), np.empty((k, nw), dtype="<u4")
        for b, it in enumerate(items):
            shift[b] = it[1][3]
        t = e.upload_words(three)
        plain = AlicePlain(None, None, None, e.upload_u64([it[1][2] & 0xFFFFFFFFFFFFFFFF for it in items]), e.upload_words(shift))
        res = e.download_words(Initiator.step_6_7_batch(e.upload_u64([it[2] for it in items]), t[2], t[0], t[1], plain, l, pai))
        return PaillierCiphertext.rows(res, pai)

    async def perform_secure_comparison_batch(self, x_enc: torch.Tensor, y_enc: torch.Tensor, draws=None,
                                              source: str = "device", engine=None, generator=None, chunks: int = 1) -> torch.Tensor:
        """B comparisons at once over the same four message exchanges, with batches on the wire as whole arrays
        (wire.py: the device arrays themselves when the transport's endpoints share a GPU, one pinned byte buffer otherwise).
        `draws` (batch.BatchDraws; Alice's fields) injects the randomness; otherwise every draw of SC/initiator.py:250, :420,
        :512, :223 and the 1 + (l+1) randomizer inputs per comparison (:205-210 scaled by B) are generated ON THE DEVICE by the
        engine's CSPRNG (`source="device"`; "torch" + `generator` = seeded, for reproducible runs) and consumed by the same fused
        launches as injected draws.  x_enc, y_enc: [B][2nw] Paillier ciphertexts.
        chunks > 1: the batch travels as that many sub-sessions on the same connection (message ids `.._chunk_i`), announced by a
        plan message; while one chunk's messages drain over the wire (copies on a stream of their own, wire.outgoing_async) the
        event loop runs the other chunks' steps -- the pipelining a transport between two processes needs."""
        from . import wire
        from .batch import draw_alice, split_draws
        from .distributed import shard_bounds

        if self.communicator is None:
            raise ValueError("Communicator not properly initialized.")
        comm = self.communicator
        self.session_id += 1
        sid = self.session_id
        got_p, got_d = wire.unpack_public_schemes(await comm.recv(self.other_party, msg_id=f"schemes_batch_session_{sid}"), engine)
        if self._scheme_paillier is None:
            self._scheme_paillier = got_p
        elif self._scheme_paillier != got_p:
            raise ValueError("Readily available Paillier scheme and received Paillier scheme are different.")
        if self._scheme_dgk is None:
            self._scheme_dgk = got_d
        elif self._scheme_dgk != got_d:
            raise ValueError("Readily available DGK scheme and received DGK scheme are different.")
        pai = self.scheme_paillier
        count = x_enc.shape[0]
        wire.expect_array(x_enc, (count, 2 * pai.mod_n.nwords), "x_enc")
        wire.expect_array(y_enc, (count, 2 * pai.mod_n.nwords), "y_enc")
        chunks = max(1, min(int(chunks), count))
        if chunks == 1:
            return await self._batch_session(f"session_{sid}", x_enc, y_enc, draws, source, generator, None)
        bounds = [shard_bounds(count, i, chunks) for i in range(chunks)]
        await comm.send(self.other_party, wire.pack_plan([b - a for a, b in bounds]), msg_id=f"step_1_batch_session_{sid}")
        parts = [None] * chunks if draws is None else split_draws(draws, bounds)
        out = torch.empty_like(x_enc)
        await asyncio.gather(*(self._batch_session(f"session_{sid}_chunk_{i}", x_enc[a:b], y_enc[a:b], parts[i], source, generator, out[a:b])
                               for i, (a, b) in enumerate(bounds)))
        return out

    async def _batch_session(self, tag: str, x_enc: torch.Tensor, y_enc: torch.Tensor, draws, source: str, generator, out) -> torch.Tensor:
        """One (sub-)session of the batched protocol: Alice's steps around the four message exchanges, message ids `.._{tag}`."""
        from . import wire
        from .batch import draw_alice

        comm, pai, dgk, l = self.communicator, self.scheme_paillier, self.scheme_dgk, self.l_maximum_bit_length
        dev, count = x_enc.device, x_enc.shape[0]
        nw_p, nw_d = pai.mod_n.nwords, dgk.mod_n.nwords
        if draws is None:
            draws = draw_alice(count, l, pai, dgk, source, generator)
        # a byte transport's messages are laid out in pinned memory up front and written by the steps' own last launches
        # (wire.reserve: no device-to-host copy afterwards); a device transport gets the arrays themselves
        msg = wire.reserve(comm, dev, (count, 2 * nw_p))
        z_enc, plain = Initiator.step_1_batch(x_enc, y_enc, l, pai, draws.r, draws.rho_z, out=None if msg is None else msg.arrays[0])
        await comm.send(self.other_party, wire.outgoing(comm, z_enc) if msg is None else await msg.finish(), msg_id=f"step_1_batch_{tag}")
        got = wire.incoming(await comm.recv(self.other_party, msg_id=f"step_4b_batch_{tag}"), dev, expect=None, planes_of_one=True)
        if len(got) == 1:            # [d] and the planes [beta_i] as ONE array, the way the key holder's launch stored them
            planes = wire.expect_array(got[0], (l + 1, count, nw_d), "[d], [beta_i]")
            d_enc, beta_enc = planes[0], planes[1:]
        elif len(got) == 2:
            d_enc = wire.expect_array(got[0], (count, nw_d), "[d]")                      # sizes come from this side's l and B,
            beta_enc = wire.expect_array(got[1], (l, count, nw_d), "[beta_i]")           # never from the message
        else:
            raise ValueError(f"batch message carries {len(got)} arrays, expected [d] and [beta_i]")
        if draws.permutation is not None and not bool(Initiator.permutation_is_valid(draws.permutation)):
            raise ValueError("permutation: a row is not a permutation of the l + 1 positions")   # before anything is computed or sent
        msg = wire.reserve(comm, dev, (l + 1, count, nw_d))
        c, _ = Initiator.step_4_batch(d_enc, beta_enc, plain, draws.delta_a, dgk, draws.rhos, draws.permutation, draws.r_alice_dgk,
                                      out=None if msg is None else msg.arrays[0])
        await comm.send(self.other_party, wire.outgoing(comm, c) if msg is None else await msg.finish(), msg_id=f"step_4i_batch_{tag}")
        got = wire.incoming(await comm.recv(self.other_party, msg_id=f"step_5_batch_{tag}"), dev, expect=None)
        if len(got) == 1:            # [[zeta_1]] | [[zeta_2]] | [[delta_B]] as the row blocks of one array
            three = wire.expect_array(got[0], (3, count, 2 * nw_p), "[[zeta_1]], [[zeta_2]], [[delta_B]]")
            zeta_1, zeta_2, delta_b_enc = three[0], three[1], three[2]
        elif len(got) == 3:
            zeta_1, zeta_2, delta_b_enc = (wire.expect_array(t, (count, 2 * nw_p), name) for t, name in
                                           ((got[0], "[[zeta_1]]"), (got[1], "[[zeta_2]]"), (got[2], "[[delta_B]]")))
        else:
            raise ValueError(f"batch message carries {len(got)} arrays, expected three ciphertext blocks")
        return Initiator.step_6_7_batch(draws.delta_a, delta_b_enc, zeta_1, zeta_2, plain, l, pai, out)   # one inversion pass

    async def receive_encryption_schemes(self, session_id: int = 1) -> None:
        """Receive Bob's public schemes; a pre-set scheme must match (SC/initiator.py:177-203)."""
        if self.communicator is None:
            raise ValueError("Communicator not properly initialized.")
        got_paillier, got_dgk = await self.communicator.recv(self.other_party, msg_id=f"schemes_session_{session_id}")
        if self._scheme_paillier is None:
            self._scheme_paillier = got_paillier
        elif self._scheme_paillier != got_paillier:
            raise ValueError("Readily available Paillier scheme and received Paillier scheme are different.")
        if self._scheme_dgk is None:
            self._scheme_dgk = got_dgk
        elif self._scheme_dgk != got_dgk:
            raise ValueError("Readily available DGK scheme and received DGK scheme are different.")

    def _start_randomness_generation(self) -> None:
        """1 Paillier + (l+1) DGK randomizers, exactly what one comparison consumes (SC/initiator.py:205-210)."""
        self.scheme_paillier.boot_randomness_generation(1)
        self.scheme_dgk.boot_randomness_generation(self.l_maximum_bit_length + 1)

    @staticmethod
    def shuffle(values: list[Any], permutation: Sequence[int] | None = None) -> list[Any]:
        """Random reordering (SC/initiator.py:212-226); `permutation[k]` = source index of output k when injected."""
        if permutation is None:
            permutation = list(range(len(values)))
            for k in range(len(permutation) - 1, 0, -1):  # Fisher-Yates on crypto randomness
                j = secrets.randbelow(k + 1)
                permutation[k], permutation[j] = permutation[j], permutation[k]
        return [values[k] for k in permutation]

    # ------------------------------------------------------------------ single-ciphertext steps
    @staticmethod
    def step_1(x_enc: PaillierCiphertext, y_enc: PaillierCiphertext, l: int, scheme_paillier: Paillier,
               r: int | None = None) -> tuple[PaillierCiphertext, int]:
        """[[z]] = [[y - x + 2^l + r]] (SC/initiator.py:228-258)."""
        n = scheme_paillier.public_key.n
        assert (1 << (l + 2)) < n // 2
        if r is None:
            r = secrets.randbelow(n)
        blind = scheme_paillier.unsafe_encrypt((1 << l) + r, apply_encoding=False)
        return y_enc - x_enc + blind, r

    @staticmethod
    def step_3(r: int, l: int) -> list[int]:
        """alpha = r mod 2^l as bits (SC/initiator.py:260-270)."""
        return to_bits(r % (1 << l), l)

    @staticmethod
    def step_4c(d_enc: DGKCiphertext, r: int, scheme_dgk: DGK, scheme_paillier: Paillier) -> DGKCiphertext:
        """[d] <- [0] when r < (N-1)/2 (SC/initiator.py:272-291)."""
        n = scheme_paillier.public_key.n
        assert 0 <= r < n
        return scheme_dgk.unsafe_encrypt(0, apply_encoding=False) if r < (n - 1) // 2 else d_enc

    @staticmethod
    def step_4d(alpha: list[int], beta_is_enc: list[DGKCiphertext]) -> list[DGKCiphertext]:
        """[alpha_i xor beta_i] (SC/initiator.py:293-328)."""
        return [b_enc if a_i == 0 else 1 - b_enc for a_i, b_enc in zip(alpha, beta_is_enc)]

    @staticmethod
    def step_4e(r: int, alpha: list[int], alpha_is_xor_beta_is_enc: list[DGKCiphertext], d_enc: DGKCiphertext,
                scheme_paillier: Paillier) -> tuple[list[DGKCiphertext], list[int]]:
        """alpha~ and the carry-corrected [w_i] (SC/initiator.py:330-384)."""
        l = len(alpha_is_xor_beta_is_enc)
        alpha_tilde = to_bits((r - scheme_paillier.public_key.n) % (1 << l), l)
        w = [x if a == at else x - d_enc for a, at, x in zip(alpha, alpha_tilde, alpha_is_xor_beta_is_enc)]
        return w, alpha_tilde

    @staticmethod
    def step_4f(w_is_enc: list[DGKCiphertext]) -> list[DGKCiphertext]:
        """[w_i] <- [w_i]^(2^i) (SC/initiator.py:386-410)."""
        return [w * (1 << i) for i, w in enumerate(w_is_enc)]

    @staticmethod
    def step_4g(delta_a: int | None = None) -> tuple[int, int]:
        """delta_A random bit, s = 1 - 2 delta_A (SC/initiator.py:412-421)."""
        if delta_a is None:
            delta_a = secrets.randbelow(2)
        return 1 - 2 * delta_a, delta_a

    @staticmethod
    def step_4h(s: int, alpha: list[int], alpha_tilde: list[int], d_enc: DGKCiphertext, beta_is_enc: list[DGKCiphertext],
                w_is_enc: list[DGKCiphertext], delta_a: int, scheme_dgk: DGK) -> list[DGKCiphertext]:
        """[c_-1], [c_0], ..., [c_{l-1}] (SC/initiator.py:423-485)."""
        l = len(beta_is_enc)
        d_pow = {-1: d_enc * -1, 0: d_enc * 0, 1: d_enc * 1}
        out: list[DGKCiphertext | None] = [None] * l
        w_sum: int | DGKCiphertext = 0
        for i in reversed(range(l)):
            c_i = scheme_dgk.unsafe_encrypt(s, apply_encoding=False)
            c_i += int(alpha[i]) + d_pow[alpha_tilde[i] - alpha[i]] - beta_is_enc[i] + 3 * w_sum
            out[i] = c_i
            w_sum += w_is_enc[i]
        head = cast(DGKCiphertext, delta_a + w_sum)
        return [head] + cast(list, out)

    @staticmethod
    def step_4i(c_is_enc: list[DGKCiphertext], scheme_dgk: DGK, do_shuffle: bool = True, rhos: Sequence[int] | None = None,
                permutation: Sequence[int] | None = None) -> list[DGKCiphertext]:
        """Blind every c_i with a random exponent in [1, u) and shuffle (SC/initiator.py:487-516)."""
        u = scheme_dgk.public_key.u
        if rhos is None:
            rhos = [secrets.randbelow(u - 1) + 1 for _ in c_is_enc]
        masked = [c * int(rho) for c, rho in zip(c_is_enc, rhos)]
        return Initiator.shuffle(masked, permutation) if do_shuffle else masked

    @staticmethod
    def step_6(delta_a: int, delta_b_enc: PaillierCiphertext) -> PaillierCiphertext:
        """[[beta < alpha]] (SC/initiator.py:518-531)."""
        return delta_b_enc if delta_a == 1 else 1 - delta_b_enc

    @staticmethod
    def step_7(zeta_1_enc: PaillierCiphertext, zeta_2_enc: PaillierCiphertext, r: int, l: int,
               beta_lt_alpha_enc: PaillierCiphertext, scheme_paillier: Paillier) -> PaillierCiphertext:
        """[[x <= y]] (SC/initiator.py:533-564)."""
        zeta_enc = zeta_1_enc if r < (scheme_paillier.public_key.n - 1) // 2 else zeta_2_enc
        return zeta_enc - (scheme_paillier.unsafe_encrypt(r >> l, apply_encoding=False) + beta_lt_alpha_enc)

    # ------------------------------------------------------------------ batched steps (device arrays): one library call each
    @staticmethod
    def step_1_batch(x_enc: torch.Tensor, y_enc: torch.Tensor, l: int, scheme_paillier: Paillier, r: torch.Tensor,
                     rho_z: torch.Tensor | None = None, randomizers_ready: bool = False,
                     out: torch.Tensor | None = None) -> tuple[torch.Tensor, AlicePlain]:
        """B times step 1 + step 3 + the plaintext side of 4c/4e/7 (sc_initiator_step1).  x_enc, y_enc: [B][2nw]; r: [B][nw]
        (injected); rho_z: [B][nw] = the `.randomize()` of [[z]] (SC/initiator.py:109) fused in -- or, with
        `randomizers_ready`, the finished randomizers rho_z^N mod N^2 ([B][2nw]) computed ahead of time."""
        n = scheme_paillier.public_key.n
        assert (1 << (l + 2)) < n // 2
        z, alpha, alpha_tilde, r_small, r_shift = scheme_paillier.engine.initiator_step1(scheme_paillier.key, l, x_enc, y_enc, r, rho_z,
                                                                                         randomizers_ready, out)
        return z, AlicePlain(r, alpha, alpha_tilde, r_small, r_shift)

    @staticmethod
    def step_4_batch(d_enc: torch.Tensor, beta_is_enc: torch.Tensor, plain: AlicePlain, delta_a: torch.Tensor, scheme_dgk: DGK,
                     rhos: torch.Tensor, permutation: torch.Tensor | None = None, randomizer_exponents: torch.Tensor | None = None,
                     want_unblinded: bool = False, randomizers_ready: bool = False,
                     out: torch.Tensor | None = None) -> tuple[torch.Tensor, torch.Tensor | None]:
        """Steps 4c .. 4i for B comparisons in ONE library call (sc_initiator_step4): the inversion pass over [d], [beta_i], the
        fused steps 4c-4h, the blinding c_i^rho_i, the re-randomization * h^r_i (`randomizer_exponents`) and the shuffle.
        With `randomizers_ready`, `randomizer_exponents` holds the finished h^r_i ([l+1][B][nw]) instead of the exponents.
        Returns ([c_i] as sent: [l+1][B][nw], and the unblinded vector of step 4h when `want_unblinded`)."""
        l = beta_is_enc.shape[0]
        return scheme_dgk.engine.initiator_step4(scheme_dgk.key, l, d_enc, beta_is_enc, plain.alpha, plain.alpha_tilde, plain.r_small, delta_a,
                                                 rhos, permutation, randomizer_exponents, want_unblinded, randomizers_ready, out)

    @staticmethod
    def step_4c_to_4h_batch(d_enc: torch.Tensor, beta_is_enc: torch.Tensor, plain: AlicePlain, delta_a: torch.Tensor,
                            scheme_dgk: DGK) -> torch.Tensor:
        """Steps 4c, 4d, 4e, 4f, 4h fused for B comparisons.  beta_is_enc: [l][B][nw] bit-major; d_enc: [B][nw];
        delta_a: [B] u64 (step 4g's draw, injected).  Returns [l+1][B][nw] = c_-1, c_0, .., c_{l-1} (not blinded)."""
        l = beta_is_enc.shape[0]
        return scheme_dgk.engine.initiator_step4(scheme_dgk.key, l, d_enc, beta_is_enc, plain.alpha, plain.alpha_tilde, plain.r_small, delta_a)[0]

    @staticmethod
    def step_4i_batch(c_is_enc: torch.Tensor, scheme_dgk: DGK, rhos: torch.Tensor, permutation: torch.Tensor | None = None,
                      randomizer_exponents: torch.Tensor | None = None) -> torch.Tensor:
        """Blinding c_i^rho_i (and, when `randomizer_exponents` is given, the `.randomize()` of SC/initiator.py:153-154
        fused in: * h^r_i), then the per-comparison shuffle -- in the store of the same launch (sc_initiator_step4i).
        c_is_enc: [l+1][B][nw]; rhos: [l+1][B][ew]; permutation: [B][l+1] int64 (output k of comparison b takes blinded c at index
        permutation[b][k]).  A row that is not a permutation of range(l+1) is replaced by the identity inside the library (every
        output row is written, nothing stale, no other comparison touched) and CHECKED lazily: permutation_is_valid is a device
        flag the caller reads where it synchronises anyway -- the interactive protocol refuses such input before anything is sent."""
        lp1, count, nw = c_is_enc.shape
        if permutation is not None and (tuple(permutation.shape) != (count, lp1) or permutation.dtype != torch.int64):
            raise ValueError(f"permutation: expected int64 [{count}][{lp1}], got {permutation.dtype} {tuple(permutation.shape)}")
        return scheme_dgk.engine.initiator_step4i(scheme_dgk.key, lp1 - 1, c_is_enc, rhos, permutation, randomizer_exponents)

    @staticmethod
    def permutation_is_valid(permutation: torch.Tensor) -> torch.Tensor:
        """Device flag (0-dim bool): every row of `permutation` ([B][k] int64) is a permutation of range(k).  Reading it
        synchronises; step_4i_batch itself never does."""
        k = permutation.shape[1]
        ref = torch.arange(k, device=permutation.device, dtype=torch.int64)
        return (torch.sort(permutation, dim=1).values == ref).all()

    @staticmethod
    def step_6_batch(delta_a: torch.Tensor, delta_b_enc: torch.Tensor, scheme_paillier: Paillier) -> torch.Tensor:
        """[[beta < alpha]] for B comparisons: delta_b_enc where delta_a == 1 else [[1]] * delta_b_enc^-1."""
        e = scheme_paillier.engine
        flipped = e.modmul_const(scheme_paillier.mod_n2, scheme_paillier.neg_batch(delta_b_enc), scheme_paillier.public_key.n + 1)
        return torch.where((delta_a != 0).reshape(-1, 1), delta_b_enc, flipped).contiguous()

    @staticmethod
    def step_7_batch(zeta_1_enc: torch.Tensor, zeta_2_enc: torch.Tensor, plain: AlicePlain, l: int,
                     beta_lt_alpha_enc: torch.Tensor, scheme_paillier: Paillier) -> torch.Tensor:
        """[[x <= y]] for B comparisons."""
        zeta = torch.where((plain.r_small != 0).reshape(-1, 1), zeta_1_enc, zeta_2_enc).contiguous()
        t = scheme_paillier.add_batch(scheme_paillier.encrypt_raw_batch(plain.r_shift), beta_lt_alpha_enc)
        return scheme_paillier.add_batch(zeta, scheme_paillier.neg_batch(t))

    @staticmethod
    def step_6_7_batch(delta_a: torch.Tensor, delta_b_enc: torch.Tensor, zeta_1_enc: torch.Tensor, zeta_2_enc: torch.Tensor,
                       plain: AlicePlain, l: int, scheme_paillier: Paillier, out: torch.Tensor | None = None) -> torch.Tensor:
        """Steps 6 and 7 in one library call (sc_initiator_step67) with ONE inversion pass instead of two, yielding the same
        residues: [[x<=y]] = [[zeta]] * ([[r div 2^l]] * [[beta<alpha]])^-1 with [[beta<alpha]] = [[delta_B]] (delta_A = 1) or
        [[1]] [[delta_B]]^-1 (delta_A = 0) equals [[zeta]] * D * [[-(r div 2^l) - (1 - delta_A)]] with D = [[delta_B]]^-1
        (delta_A = 1) or [[delta_B]] (delta_A = 0), because [[a]] [[b]] = [[a + b]] holds exactly for unrandomized
        g = N + 1 encryptions (SC/initiator.py:529-531, 558-563).  `out`: write the results into this [B][2nw] array (a shard's
        row block of the whole batch's result) instead of a new one."""
        return scheme_paillier.engine.initiator_step67(scheme_paillier.key, delta_a, delta_b_enc, zeta_1_enc, zeta_2_enc, plain.r_small,
                                                       plain.r_shift, out)
