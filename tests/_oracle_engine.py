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

    # ---- plumbing
    def upload(self, xs, nwords):
        return torch.from_numpy(ints_to_words(xs, nwords).view(np.int32))

    def upload_u64(self, xs):
        return torch.from_numpy(np.array(list(xs), dtype=np.uint64).view(np.int64))

    def download(self, t):
        return words_to_ints(t.detach().cpu().contiguous().numpy().view(np.uint32))

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
