"""Python face of the C ABI: device-resident residue batches (torch tensors as plain device memory).

All arithmetic happens in libsc_amd.so on the GPU; this module only moves pointers around.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Iterable

import numpy as np
import torch

from . import _lib
from .limbs import int_to_words, ints_to_words, words_to_ints

ScError = _lib.ScError


class NotInvertibleError(ZeroDivisionError):
    """An element of a batch has no modular inverse (the reference's pow/gmpy2 raise likewise)."""


@dataclass(frozen=True)
class Modulus:
    id: int
    n: int
    nwords: int


@dataclass(frozen=True)
class FixedBase:
    id: int
    mod: Modulus
    exp_bits: int
    window: int


class Engine:
    """One context per process/device.  Tensors are int32 views of uint32 words, shape [count, nwords]."""

    def __init__(self, device: int | None = None) -> None:
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise ScError("no GPU visible: the secure-comparison engine has no CPU fallback")
        self.device_index = torch.cuda.current_device() if device is None else device
        self.device = torch.device("cuda", self.device_index)
        ctx = C.c_void_p()
        rc = self.lib.sc_ctx_create(self.device_index, C.byref(ctx))
        if rc != 0:
            raise ScError(f"sc_ctx_create failed ({rc})")
        self.ctx = ctx
        self._mods: dict[tuple[int, int], Modulus] = {}
        self._exps: dict[int, int] = {}
        self._consts: dict[tuple[int, int], int] = {}

    def close(self) -> None:
        if getattr(self, "ctx", None):
            self.lib.sc_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _check(self, rc: int) -> None:
        if rc == 0:
            return
        msg = self.lib.sc_last_error(self.ctx).decode()
        if rc == -1:
            raise ValueError(msg)
        if rc == -3:
            raise NotInvertibleError(msg)
        raise ScError(f"{msg} (status {rc})")

    def _sync_stream(self) -> None:
        self.lib.sc_ctx_set_stream(self.ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))

    @staticmethod
    def _ptr(t: torch.Tensor | None) -> C.c_void_p:
        return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())

    def _host_words(self, x: int, nwords: int):
        arr = int_to_words(x, nwords)
        return arr, arr.ctypes.data_as(C.c_void_p)

    def upload(self, xs: Iterable[int], nwords: int) -> torch.Tensor:
        arr = ints_to_words(xs, nwords)
        return torch.from_numpy(arr.view(np.int32)).to(self.device)

    def upload_u64(self, xs: Iterable[int]) -> torch.Tensor:
        arr = np.array(list(xs), dtype=np.uint64)
        return torch.from_numpy(arr.view(np.int64)).to(self.device)

    def download(self, t: torch.Tensor) -> list[int]:
        return words_to_ints(t.detach().cpu().numpy().view(np.uint32))

    def empty(self, count: int, nwords: int) -> torch.Tensor:
        return torch.empty((count, nwords), dtype=torch.int32, device=self.device)

    def synchronize(self) -> None:
        torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------ registration
    def modulus(self, n: int, nwords: int | None = None) -> Modulus:
        nwords = nwords or (n.bit_length() + 31) // 32
        key = (n, nwords)
        if key not in self._mods:
            arr, p = self._host_words(n, nwords)
            mid = C.c_int()
            self._check(self.lib.sc_mod_create(self.ctx, p, nwords, C.byref(mid)))
            self._mods[key] = Modulus(mid.value, n, nwords)
        return self._mods[key]

    def exponent(self, e: int) -> int:
        if e < 0:
            raise ValueError("negative exponent")
        if e not in self._exps:
            nw = max(1, (e.bit_length() + 31) // 32)
            arr, p = self._host_words(e, nw)
            eid = C.c_int()
            self._check(self.lib.sc_exp_create(self.ctx, p, nw, C.byref(eid)))
            self._exps[e] = eid.value
        return self._exps[e]

    def constant(self, mod: Modulus, v: int) -> int:
        key = (mod.id, v)
        if key not in self._consts:
            arr, p = self._host_words(v % mod.n, mod.nwords)
            cid = C.c_int()
            self._check(self.lib.sc_const_create(self.ctx, mod.id, p, mod.nwords, C.byref(cid)))
            self._consts[key] = cid.value
        return self._consts[key]

    def fixed_base(self, mod: Modulus, base: int, exp_bits: int, window: int = 8) -> FixedBase:
        self._sync_stream()
        arr, p = self._host_words(base % mod.n, mod.nwords)
        fid = C.c_int()
        self._check(self.lib.sc_fbt_create(self.ctx, mod.id, p, exp_bits, window, C.byref(fid)))
        return FixedBase(fid.value, mod, exp_bits, window)

    # ------------------------------------------------------------------ batched residue arithmetic
    @staticmethod
    def _stride(t: torch.Tensor, count: int) -> int:
        return 0 if (t.dim() == 1 or t.shape[0] == 1) and count != 1 else t.shape[-1]

    def modmul(self, mod: Modulus, a: torch.Tensor, b: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        count = max(a.shape[0] if a.dim() > 1 else 1, b.shape[0] if b.dim() > 1 else 1)
        out = self.empty(count, mod.nwords) if out is None else out
        self._sync_stream()
        self._check(self.lib.sc_modmul(self.ctx, mod.id, self._ptr(a), self._stride(a, count), self._ptr(b),
                                       self._stride(b, count), self._ptr(out), count))
        return out

    def modmul_const(self, mod: Modulus, a: torch.Tensor, c: int, out: torch.Tensor | None = None) -> torch.Tensor:
        count = a.shape[0]
        out = self.empty(count, mod.nwords) if out is None else out
        self._sync_stream()
        self._check(self.lib.sc_modmul_const(self.ctx, mod.id, self._ptr(a), self.constant(mod, c), self._ptr(out), count))
        return out

    def modexp_shared(self, mod: Modulus, x: torch.Tensor, e: int, mul_into: torch.Tensor | None = None,
                      out: torch.Tensor | None = None) -> torch.Tensor:
        count = x.shape[0]
        out = self.empty(count, mod.nwords) if out is None else out
        self._sync_stream()
        self._check(self.lib.sc_modexp_shared(self.ctx, mod.id, self.exponent(e), self._ptr(x), x.shape[-1],
                                              self._ptr(mul_into), self._ptr(out), count))
        return out

    def supports_sq(self, mod: Modulus) -> bool:
        """Can x^e mod m^2 be computed with products modulo m (pair arithmetic) for this modulus?"""
        return self.lib.sc_mod_supports_sq(self.ctx, mod.id) == 1

    def modexp_shared_sq(self, mod_m: Modulus, mod_m2: Modulus, x: torch.Tensor, e: int, mul_into: torch.Tensor | None = None,
                         out: torch.Tensor | None = None) -> torch.Tensor:
        """x^e mod m^2 [* mul_into] via pair arithmetic modulo m (identical residues, ~0.6x the multiply-adds)."""
        count = x.shape[0]
        out = self.empty(count, mod_m2.nwords) if out is None else out
        self._sync_stream()
        self._check(self.lib.sc_modexp_shared_sq(self.ctx, mod_m.id, mod_m2.id, self.exponent(e), self._ptr(x), x.shape[-1],
                                                 self._ptr(mul_into), self._ptr(out), count))
        return out

    def modexp_shared_isone(self, mod: Modulus, x: torch.Tensor, e: int) -> torch.Tensor:
        count = x.shape[0]
        flags = torch.empty((count,), dtype=torch.uint8, device=self.device)
        self._sync_stream()
        self._check(self.lib.sc_modexp_shared_isone(self.ctx, mod.id, self.exponent(e), self._ptr(x), x.shape[-1],
                                                    self._ptr(flags), count))
        return flags

    def fixedbase_pow(self, fb: FixedBase, e: torch.Tensor, mul_into: torch.Tensor | None = None,
                      out: torch.Tensor | None = None) -> torch.Tensor:
        count = e.shape[0]
        out = self.empty(count, fb.mod.nwords) if out is None else out
        self._sync_stream()
        self._check(self.lib.sc_fixedbase_pow(self.ctx, fb.id, self._ptr(e), e.shape[-1], self._ptr(mul_into),
                                              self._ptr(out), count))
        return out

    def modexp_var(self, mod: Modulus, x: torch.Tensor, e: torch.Tensor, ebits: int, fb: FixedBase | None = None,
                   e2: torch.Tensor | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
        count = x.shape[0]
        out = self.empty(count, mod.nwords) if out is None else out
        self._sync_stream()
        self._check(self.lib.sc_modexp_var(self.ctx, mod.id, self._ptr(x), self._ptr(e), e.shape[-1], ebits,
                                           -1 if fb is None else fb.id, self._ptr(e2), 0 if e2 is None else e2.shape[-1],
                                           self._ptr(out), count))
        return out

    def modinv(self, mod: Modulus, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        count = x.shape[0]
        out = self.empty(count, mod.nwords) if out is None else out
        self._sync_stream()
        bad = C.c_int64(-1)
        self._check(self.lib.sc_modinv(self.ctx, mod.id, self._ptr(x), self._ptr(out), count, C.byref(bad)))
        return out

    def paillier_encrypt_raw(self, mod_n2: Modulus, n: int, m: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        count = m.shape[0]
        out = self.empty(count, mod_n2.nwords) if out is None else out
        self._sync_stream()
        self._check(self.lib.sc_paillier_encrypt_raw(self.ctx, mod_n2.id, self.constant(mod_n2, n), self._ptr(m),
                                                     m.shape[-1], self._ptr(out), count))
        return out

    def paillier_encrypt_raw_neg(self, mod_n2: Modulus, n: int, m: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """[[-m]] = 1 - m N mod N^2 (the inverse of paillier_encrypt_raw's result, without an inversion)."""
        count = m.shape[0]
        out = self.empty(count, mod_n2.nwords) if out is None else out
        self._sync_stream()
        self._check(self.lib.sc_paillier_encrypt_raw_neg(self.ctx, mod_n2.id, self.constant(mod_n2, n), self._ptr(m),
                                                         m.shape[-1], self._ptr(out), count))
        return out

    def paillier_l_mul(self, mod: Modulus, k: int, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        count = x.shape[0]
        out = self.empty(count, mod.nwords) if out is None else out
        self._sync_stream()
        self._check(self.lib.sc_paillier_l_mul(self.ctx, mod.id, self.constant(mod, k), self._ptr(x), x.shape[-1],
                                               self._ptr(out), count))
        return out

    def crt_combine(self, mod_p: Modulus, mod_full: Modulus, mq: int, a_p: torch.Tensor, a_q: torch.Tensor,
                    out: torch.Tensor | None = None) -> torch.Tensor:
        """x with x = a_p (mod m_p), x = a_q (mod m_q), m_p m_q = mod_full.n (CRT recombination on the GPU)."""
        count = a_p.shape[0]
        out = self.empty(count, mod_full.nwords) if out is None else out
        k = pow(mq, -1, mod_p.n)
        self._sync_stream()
        self._check(self.lib.sc_crt_combine(self.ctx, mod_p.id, mod_full.id, self.constant(mod_p, k), self.constant(mod_p, mod_p.n - k),
                                            self.constant(mod_full, mq), self._ptr(a_p), a_p.shape[-1], self._ptr(a_q), a_q.shape[-1],
                                            self._ptr(out), count))
        return out

    def plain_alice(self, r: torch.Tensor, n: int, l: int):
        count, nw = r.shape
        m1 = self.empty(count, nw + 1)
        alpha = torch.empty((count,), dtype=torch.int64, device=self.device)
        alpha_t = torch.empty_like(alpha)
        rsmall = torch.empty_like(alpha)
        rshift = self.empty(count, nw)
        arr, p = self._host_words(n, nw)
        self._sync_stream()
        self._check(self.lib.sc_plain_alice(self.ctx, self._ptr(r), p, nw, l, count, self._ptr(m1), self._ptr(alpha),
                                            self._ptr(alpha_t), self._ptr(rsmall), self._ptr(rshift)))
        return m1, alpha, alpha_t, rsmall, rshift

    def plain_bob(self, z: torch.Tensor, n: int, l: int):
        count, nw = z.shape
        beta = torch.empty((count,), dtype=torch.int64, device=self.device)
        dbit = torch.empty_like(beta)
        zeta1 = self.empty(count, nw)
        zeta2 = self.empty(count, nw)
        arr, p = self._host_words(n, nw)
        self._sync_stream()
        self._check(self.lib.sc_plain_bob(self.ctx, self._ptr(z), p, nw, l, count, self._ptr(beta), self._ptr(dbit),
                                          self._ptr(zeta1), self._ptr(zeta2)))
        return beta, dbit, zeta1, zeta2

    def dgk_step4(self, mod: Modulus, g: int, g_inv: int, l: int, beta: torch.Tensor, beta_inv: torch.Tensor,
                  d: torch.Tensor, d_inv: torch.Tensor, alpha: torch.Tensor, alpha_tilde: torch.Tensor,
                  rsmall: torch.Tensor, delta_a: torch.Tensor) -> torch.Tensor:
        count = d.shape[0]
        out = torch.empty((l + 1, count, mod.nwords), dtype=torch.int32, device=self.device)
        self._sync_stream()
        self._check(self.lib.sc_dgk_step4(self.ctx, mod.id, self.constant(mod, g), self.constant(mod, g_inv), l,
                                          self._ptr(beta), self._ptr(beta_inv), self._ptr(d), self._ptr(d_inv),
                                          self._ptr(alpha), self._ptr(alpha_tilde), self._ptr(rsmall), self._ptr(delta_a),
                                          self._ptr(out), count))
        return out

    def peak_probe(self) -> float:
        v = C.c_double()
        self._sync_stream()
        self._check(self.lib.sc_peak_probe(self.ctx, C.byref(v)))
        return v.value

    def set_latency_mode(self, mode: int) -> None:
        """Small-batch kernel policy (sc_ctx_set_latency_mode): 0 never, 1 automatic (default), 2 whenever available."""
        self._check(self.lib.sc_ctx_set_latency_mode(self.ctx, int(mode)))

    def table_traffic_probe(self, mod: Modulus, x: torch.Tensor, entries: int, reads: int) -> tuple[torch.Tensor, int]:
        """Measurement aid (sc_table_traffic_probe): returns (x again, limbs per table row)."""
        count = x.shape[0]
        out = self.empty(count, mod.nwords)
        s = C.c_int()
        self._sync_stream()
        self._check(self.lib.sc_table_traffic_probe(self.ctx, mod.id, self._ptr(x), self._ptr(out), count, entries, reads, C.byref(s)))
        return out, s.value

    def mac_counter(self, reset: bool = False) -> float:
        v = C.c_double()
        self._check(self.lib.sc_mac_counter(self.ctx, int(reset), C.byref(v)))
        return v.value
