"""TEST-ONLY stand-in for protocols.secure_comparison_amd.engine.Engine that does the arithmetic with the CPU
oracle's Python ints on CPU torch tensors.  It lets the `-m "not gpu"` suite exercise the host logic (step
choreography, bit-major layouts, sharding) without a GPU.  It lives under tests/ on purpose: the product never
imports it, and the GPU parity tests never use it."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from oracle import sc_oracle as o
from protocols.secure_comparison_amd.limbs import ints_to_words, words_to_ints


@dataclass(frozen=True)
class Modulus:
    id: int
    n: int
    nwords: int


@dataclass(frozen=True)
class FixedBase:
    id: int
    mod: Modulus
    base: int


@dataclass(frozen=True)
class PaillierKey:
    id: int
    mod_n: Modulus
    mod_n2: Modulus
    secret: bool
    sk: object = None          # oracle PaillierKey (test stand-in only)


@dataclass(frozen=True)
class DgkKey:
    id: int
    mod_n: Modulus
    mod_p: object
    u: int
    randomizer_bits: int
    window: int
    secret: bool
    sk: object = None          # oracle DGKKey


class OracleEngine:
    device = torch.device("cpu")

    def __init__(self):
        self._mods = {}
        self._rng_key, self._rng_call = None, 0

    # ---- the library's generator, restated (oracle/chacha_rng.py)
    def rng_seed(self, key=None):
        import os

        self._rng_key, self._rng_call = (os.urandom(32) if key is None else bytes(key)), 0

    def _rng_next_call(self):
        if self._rng_key is None:
            self.rng_seed()
        self._rng_call += 1
        return self._rng_call - 1

    def rng_bits(self, bits, count):
        from oracle import chacha_rng as c

        return self.upload(c.rng_bits(self._rng_key, self._rng_next_call(), bits, count) if count else [], (bits + 31) // 32) \
            if count else self.empty(0, (bits + 31) // 32)

    def rng_below(self, n, count, nonzero=False):
        from oracle import chacha_rng as c

        call = self._rng_next_call()
        return self.upload(c.rng_below(self._rng_key, call, n, count, nonzero), (n.bit_length() + 31) // 32)

    def rng_coins(self, count):
        from oracle import chacha_rng as c

        call = self._rng_next_call()
        return torch.tensor(c.rng_coins(self._rng_key, call, count), dtype=torch.int64)

    def rng_permutations(self, k, count):
        from oracle import chacha_rng as c

        call = self._rng_next_call()
        return torch.tensor(c.rng_permutations(self._rng_key, call, k, count), dtype=torch.int64).reshape(count, k)

    # ---- scheme-level entry points (the library composes these from its kernels; here: the oracle's integers)
    def paillier_key(self, n, p=None, q=None, use_crt=True, use_pairs=True):
        nw = (n.bit_length() + 31) // 32
        return PaillierKey(0, self.modulus(n, nw), self.modulus(n * n, 2 * nw), p is not None, o.PaillierKey(n, p, q))

    def paillier_encrypt(self, key, m, negate=False, out=None):
        n, n2 = key.mod_n.n, key.mod_n2.n
        return self.upload([(1 + (-(v % n) if negate else (v % n)) * n) % n2 for v in self._ints(m)], key.mod_n2.nwords)

    def paillier_randomize(self, key, c, rho, out=None):
        n, n2 = key.mod_n.n, key.mod_n2.n
        r = [pow(v, n, n2) for v in self._ints(rho)]
        if c is not None:
            r = [a * b % n2 for a, b in zip(self._ints(c), r)]
        return self.upload(r, key.mod_n2.nwords)

    def paillier_decrypt(self, key, c):
        return self.upload([key.sk.dec_raw(v) for v in self._ints(c)], key.mod_n.nwords)

    def dgk_key(self, n, g, h, u, t, p=None, q=None, v_p=None, v_q=None, randomizer_bits=400, window=8, use_crt=True, table_source=None):
        nw = (n.bit_length() + 31) // 32
        sk = o.DGKKey(n, g, h, u, t, p, q, v_p, v_q)
        return DgkKey(0, self.modulus(n, nw), None if p is None else self.modulus(p), u, randomizer_bits, window, p is not None, sk)

    def dgk_table_bytes(self, key):
        return 1

    def dgk_randomize(self, key, c, r, out=None):
        n = key.mod_n.n
        hr = [pow(key.sk.h, v, n) for v in self._ints(r)]
        if c is not None:
            hr = [a * b % n for a, b in zip(self._ints(c), hr)]
        return self.upload(hr, key.mod_n.nwords)

    def dgk_encrypt_bits_randomized(self, key, bits, r):
        n = key.mod_n.n
        return self.upload([(key.sk.g if b else 1) * pow(key.sk.h, v, n) % n for b, v in zip(bits.tolist(), self._ints(r))], key.mod_n.nwords)

    def dgk_is_zero(self, key, c):
        return self.modexp_shared_isone(key.mod_p, c, key.sk.v_p)

    def dgk_any_zero(self, key, c):
        planes, count, nw = c.shape
        return self.modexp_shared_isone_any(key.mod_p, c.reshape(planes * count, nw), key.sk.v_p, count)

    def initiator_step1(self, key, l, x_enc, y_enc, r, rho_z=None, ready=False, out=None):
        n, n2 = key.mod_n.n, key.mod_n2.n
        m1, alpha, alpha_t, rsmall, rshift = self.plain_alice(r, n, l)
        z = self.modmul(key.mod_n2, self.modmul(key.mod_n2, y_enc, self.modinv(key.mod_n2, x_enc)), self.paillier_encrypt(key, m1))
        if rho_z is not None:
            z = self.modmul(key.mod_n2, z, rho_z) if ready else self.paillier_randomize(key, z, rho_z)
        return z, alpha, alpha_t, rsmall, rshift

    def keyholder_step2_4b(self, pkey, dkey, l, z_enc, r_rand=None, ready=False, out=None):
        count = z_enc.shape[0]
        z = self.paillier_decrypt(pkey, z_enc)
        beta, dbit, zeta1, zeta2 = self.plain_bob(z, pkey.mod_n.n, l)
        M = (1 << 64) - 1
        bits = [int(v) & 1 for v in dbit.tolist()]
        for i in range(l):
            bits += [((int(v) & M) >> i) & 1 for v in beta.tolist()]
        bt = torch.tensor(bits, dtype=torch.uint8)
        if r_rand is not None and ready:
            enc = self.modmul_const_sel(dkey.mod_n, r_rand, None, dkey.sk.g, bt)
        elif r_rand is not None:
            enc = self.dgk_encrypt_bits_randomized(dkey, bt, r_rand)
        else:
            enc = self.upload([dkey.sk.g if b else 1 for b in bits], dkey.mod_n.nwords)
        return z, beta, dbit, zeta1, zeta2, enc.reshape(l + 1, count, -1)

    def initiator_step4i(self, key, l, c_in, rhos, permutation=None, r_rand=None, ready=False):
        lp1, count, nw = c_in.shape
        dest = None
        if permutation is not None:
            planes = torch.arange(lp1, dtype=torch.int64).expand(count, lp1)
            valid = (torch.sort(permutation, dim=1).values == planes).all(dim=1, keepdim=True)
            permutation = torch.where(valid, permutation, planes)          # a row that is not a permutation: the identity
            inverse = torch.zeros_like(permutation).scatter_(1, permutation, planes)
            dest = (inverse.t() * count + torch.arange(count, dtype=torch.int64)).reshape(-1)
        if ready and r_rand is not None:        # h^r computed ahead: blind, multiply, then place
            flat = self.modexp_var(key.mod_n, c_in.reshape(lp1 * count, nw), rhos.reshape(lp1 * count, -1), 0)
            flat = self.modmul(key.mod_n, flat, r_rand.reshape(lp1 * count, nw))
            if dest is not None:
                placed = torch.zeros_like(flat)
                placed[dest] = flat
                flat = placed
            return flat.reshape(lp1, count, nw)
        fb = FixedBase(0, key.mod_n, key.sk.h) if r_rand is not None else None
        flat = self.modexp_var(key.mod_n, c_in.reshape(lp1 * count, nw), rhos.reshape(lp1 * count, -1), 0, fb,
                               None if r_rand is None else r_rand.reshape(lp1 * count, -1), dest=dest)
        return flat.reshape(lp1, count, nw)

    def initiator_step4(self, key, l, d_enc, beta_enc, alpha, alpha_tilde, rsmall, delta_a, rhos=None, permutation=None, r_rand=None,
                        want_unblinded=False, ready=False, out=None):
        count, nw = d_enc.shape
        inv = self.modinv(key.mod_n, torch.cat([d_enc.reshape(1, count, nw), beta_enc]).reshape((l + 1) * count, nw))
        c_h = self.dgk_step4(key.mod_n, key.sk.g, o.mod_inv(key.sk.g, key.mod_n.n), l, beta_enc, inv[count:].reshape(l, count, nw), d_enc,
                             inv[:count], alpha, alpha_tilde, rsmall, delta_a)
        if rhos is None:
            return c_h, None
        return self.initiator_step4i(key, l, c_h, rhos, permutation, r_rand, ready), (c_h if want_unblinded else None)

    def keyholder_step4j_5(self, pkey, dkey, l, c_enc, zeta1, zeta2, rho3=None, ready=False, out=None):
        count = zeta1.shape[0]
        delta_b = self.dgk_any_zero(dkey, c_enc.reshape(l + 1, count, -1))
        db = self.upload([int(v) for v in delta_b.tolist()], zeta1.shape[-1])
        enc = self.paillier_encrypt(pkey, torch.cat([zeta1, zeta2, db], dim=0))
        if rho3 is not None:
            enc = self.modmul(pkey.mod_n2, enc, rho3) if ready else self.paillier_randomize(pkey, enc, rho3)
        return delta_b, enc

    def initiator_step67(self, key, delta_a, delta_b_enc, zeta1_enc, zeta2_enc, rsmall, rshift, out=None):
        # the LITERAL formulas of SC/initiator.py:529-531, 558-563 (two inversions): the library's one-inversion form must equal them
        n, n2 = key.mod_n.n, key.mod_n2.n
        res = []
        for da, db, z1, z2, rs, sh in zip(delta_a.tolist(), self._ints(delta_b_enc), self._ints(zeta1_enc), self._ints(zeta2_enc),
                                          rsmall.tolist(), self._ints(rshift)):
            blta = db if da else (1 + n) * o.mod_inv(db, n2) % n2
            zeta = z1 if rs else z2
            res.append(zeta * o.mod_inv((1 + sh * n) % n2 * blta % n2, n2) % n2)
        fresh = self.upload(res, key.mod_n2.nwords)
        if out is None:
            return fresh
        out.copy_(fresh)
        return out

    # ---- plumbing
    def upload(self, xs, nwords):
        return torch.from_numpy(ints_to_words(xs, nwords).view(np.int32))

    def upload_u64(self, xs):
        return torch.from_numpy(np.array(list(xs), dtype=np.uint64).view(np.int64))

    def download(self, t):
        return words_to_ints(t.detach().cpu().contiguous().numpy().view(np.uint32))

    def upload_words(self, arr):
        return torch.from_numpy(np.ascontiguousarray(arr, dtype="<u4").view(np.int32).copy())

    def download_words(self, t):
        return t.detach().cpu().contiguous().numpy().view(np.uint32)

    def empty(self, count, nwords):
        return torch.zeros((count, nwords), dtype=torch.int32)

    def synchronize(self):
        pass

    def modulus(self, n, nwords=None):
        nwords = nwords or (n.bit_length() + 31) // 32
        return self._mods.setdefault((n, nwords), Modulus(len(self._mods), n, nwords))

    def fixed_base(self, mod, base, exp_bits, window=8):
        return FixedBase(0, mod, base)

    def _ints(self, t, count=None):
        v = self.download(t.reshape(-1, t.shape[-1]))
        if count is not None and len(v) == 1 and count != 1:
            v = v * count
        return v

    # ---- arithmetic
    def modmul(self, mod, a, b, out=None):
        count = max(a.shape[0] if a.dim() > 1 else 1, b.shape[0] if b.dim() > 1 else 1)
        return self.upload([x * y % mod.n for x, y in zip(self._ints(a, count), self._ints(b, count))], mod.nwords)

    def modmul_const(self, mod, a, c, out=None):
        return self.upload([x * c % mod.n for x in self._ints(a)], mod.nwords)

    def modmul_const_sel(self, mod, a, c0, c1, flags, out=None):
        k0, k1 = (1 if c0 is None else c0), (1 if c1 is None else c1)
        return self.upload([x * (k1 if f else k0) % mod.n for x, f in zip(self._ints(a), flags.tolist())], mod.nwords)

    def modexp_shared(self, mod, x, e, mul_into=None, out=None):
        r = [pow(v, e, mod.n) for v in self._ints(x)]
        if mul_into is not None:
            r = [a * b % mod.n for a, b in zip(r, self._ints(mul_into))]
        return self.upload(r, mod.nwords)

    def supports_sq(self, mod):
        return True

    def modexp_shared_sq(self, mod_m, mod_m2, x, e, mul_into=None, out=None):
        return self.modexp_shared(mod_m2, x, e, mul_into, out)

    def modexp_shared_isone(self, mod, x, e):
        return torch.tensor([int(pow(v, e, mod.n) == 1) for v in self._ints(x)], dtype=torch.uint8)

    def modexp_shared_isone_any(self, mod, x, e, inner):
        flags = [int(pow(v, e, mod.n) == 1) for v in self._ints(x)]
        return torch.tensor([int(any(flags[b::inner])) for b in range(inner)], dtype=torch.int64)

    def fixedbase_pow(self, fb, e, mul_into=None, out=None):
        r = [pow(fb.base, v, fb.mod.n) for v in self._ints(e)]
        if mul_into is not None:
            r = [a * b % fb.mod.n for a, b in zip(r, self._ints(mul_into))]
        return self.upload(r, fb.mod.nwords)

    def modexp_var(self, mod, x, e, ebits, fb=None, e2=None, out=None, dest=None):
        r = [pow(a, b, mod.n) for a, b in zip(self._ints(x), self._ints(e))]
        if fb is not None:
            r = [a * pow(fb.base, b, mod.n) % mod.n for a, b in zip(r, self._ints(e2))]
        if dest is not None:
            placed = [0] * len(r)
            for i, d in enumerate(dest.tolist()):
                placed[d] = r[i]
            r = placed
        return self.upload(r, mod.nwords)

    def modinv(self, mod, x, out=None):
        return self.upload([o.mod_inv(v, mod.n) for v in self._ints(x)], mod.nwords)

    def paillier_encrypt_raw(self, mod_n2, n, m, out=None):
        return self.upload([(1 + (v % n) * n) % mod_n2.n for v in self._ints(m)], mod_n2.nwords)

    def paillier_encrypt_raw_neg(self, mod_n2, n, m, out=None):
        return self.upload([(1 - (v % n) * n) % mod_n2.n for v in self._ints(m)], mod_n2.nwords)

    def paillier_l_mul(self, mod, k, x, out=None):
        return self.upload([((v % (mod.n * mod.n)) - 1) // mod.n * k % mod.n for v in self._ints(x)], mod.nwords)

    def crt_combine(self, mod_p, mod_full, mq, a_p, a_q, out=None):
        k = pow(mq, -1, mod_p.n)
        return self.upload([(y + mq * ((x - y) * k % mod_p.n)) % mod_full.n for x, y in zip(self._ints(a_p), self._ints(a_q))], mod_full.nwords)

    def plain_alice(self, r, n, l):
        rs = self._ints(r)
        nw = r.shape[-1]
        return (self.upload([(1 << l) + v for v in rs], nw + 1), self.upload_u64([v % (1 << l) for v in rs]),
                self.upload_u64([(v - n) % (1 << l) for v in rs]), self.upload_u64([int(v < (n - 1) // 2) for v in rs]),
                self.upload([v >> l for v in rs], nw))

    def plain_bob(self, z, n, l):
        zs = self._ints(z)
        nw = z.shape[-1]
        return (self.upload_u64([v % (1 << l) for v in zs]), self.upload_u64([int(v < (n - 1) // 2) for v in zs]),
                self.upload([v >> l for v in zs], nw),
                self.upload([((v + n) >> l) if v < (n - 1) // 2 else (v >> l) for v in zs], nw))

    def dgk_step4(self, mod, g, g_inv, l, beta, beta_inv, d, d_inv, alpha, alpha_tilde, rsmall, delta_a):
        count, nw = d.shape
        n = mod.n
        dgk = o.DGKKey(n, g, 1, 0, 0)
        B = [self._ints(beta[i]) for i in range(l)]
        D = self._ints(d)
        M = (1 << 64) - 1
        out = [[None] * count for _ in range(l + 1)]
        pk_dummy = None
        for c in range(count):
            a_bits = o.to_bits((int(alpha[c]) & M) % (1 << l), l)
            at_bits = o.to_bits((int(alpha_tilde[c]) & M) % (1 << l), l)
            da = int(delta_a[c])
            d2 = dgk.enc_raw(0) if int(rsmall[c]) else D[c]
            b_enc = [B[i][c] for i in range(l)]
            xor = o.step_4d(a_bits, b_enc, dgk)
            w = [x if a == at else dgk.add(x, dgk.neg(d2)) for a, at, x in zip(a_bits, at_bits, xor)]
            w = o.step_4f(w, dgk)
            res = o.step_4h(1 - 2 * da, a_bits, at_bits, d2, b_enc, w, da, dgk)
            for i in range(l + 1):
                out[i][c] = res[i]
        return torch.stack([self.upload(row, nw) for row in out])

    def peak_probe(self):
        return 0.0

    def mac_counter(self, reset=False):
        return 0.0
