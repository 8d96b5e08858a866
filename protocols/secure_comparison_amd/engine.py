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
    """An element of a batch has no modular inverse (the reference's pow/gmpy2 raise likewise); `index` names it."""

    def __init__(self, msg: str, index: int = -1) -> None:
        super().__init__(msg)
        self.index = index


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


@dataclass(frozen=True)
class PaillierKey:
    """Handle of a library-side Paillier key object (sc_paillier_key_create)."""

    id: int
    mod_n: Modulus
    mod_n2: Modulus
    secret: bool


@dataclass(frozen=True)
class DgkKey:
    """Handle of a library-side DGK key object (sc_dgk_key_create)."""

    id: int
    mod_n: Modulus
    mod_p: Modulus | None
    u: int
    randomizer_bits: int
    window: int
    secret: bool


_HW_QUEUES_WARNED = False


def _default_hw_queues() -> None:
    """The HIP runtime multiplexes a process's streams onto 4 hardware queues by default, and streams that land on one queue run in
    sequence.  Two library contexts with their fork streams are four streams already (configs[1]: 114 k/s with a queue each, 80 k/s
    when a fifth stream of the same process made two of them share one); byte-transport sessions add copy streams, background
    randomizers another context.  The runtime reads GPU_MAX_HW_QUEUES when it initialises, i.e. at the first GPU call of the process:
    the FIRST ENGINE of a process asks for 32 -- unless the operator exported a value (it wins), and with a warning when the runtime is
    up already and the default of 4 is what it got (round 4 set the variable when the package was imported: a library changing its
    host's runtime configuration at import)."""
    global _HW_QUEUES_WARNED
    import os
    import warnings

    if "GPU_MAX_HW_QUEUES" in os.environ:
        return
    if torch.cuda.is_initialized():
        if not _HW_QUEUES_WARNED:
            _HW_QUEUES_WARNED = True
            warnings.warn("the HIP runtime was initialised before the first secure-comparison engine and GPU_MAX_HW_QUEUES is not set: streams of "
                          "concurrent contexts may share the runtime's 4 default hardware queues and run in sequence (export GPU_MAX_HW_QUEUES=32 "
                          "before the process's first GPU call)", RuntimeWarning, stacklevel=3)
        return
    os.environ["GPU_MAX_HW_QUEUES"] = "32"


class Engine:
    """One library context (sc_ctx): one per process and device, or one per concurrent shard / session thread of a device.
    Tensors are int32 views of uint32 words, shape [count, nwords].  An engine belongs to ONE host thread at a time: it orders its
    work on one stream, reuses its temporaries from call to call, and its generator numbers its calls (rng_seed(key) is for tests:
    re-seeding with the same key replays the same streams)."""

    def __init__(self, device: int | None = None) -> None:
        self.lib = _lib.load()
        _default_hw_queues()
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
        self._crt_k: dict[tuple[int, int], int] = {}          # (m_q, m_p) -> m_q^-1 mod m_p
        self._host_n: dict[tuple[int, int], tuple] = {}       # (n, nwords) -> (array, pointer) kept alive for the plain-word kernels
        self._pkeys: dict[tuple, "PaillierKey"] = {}          # library-side key objects, one per distinct key and option set:
        self._dkeys: dict[tuple, "DgkKey"] = {}               # scheme objects come and go (one per session), their keys do not

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
            raise NotInvertibleError(msg, int(self.lib.sc_last_bad_index(self.ctx)))
        raise ScError(f"{msg} (status {rc})")

    def _sync_stream(self) -> None:
        # a failed switch leaves the context on its previous stream, unordered against torch's current one: never ignore it
        self._check(self.lib.sc_ctx_set_stream(self.ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    @staticmethod
    def _ptr(t: torch.Tensor | None) -> C.c_void_p:
        return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())

    def _arr(self, t: torch.Tensor | None, name: str, rows: int | None = None, words: int | None = None, *, dtype=torch.int32,
             optional: bool = False, broadcast: bool = False) -> torch.Tensor | None:
        """The kernels read `rows * words` elements from a raw pointer: refuse anything that is not exactly that array
        (peer-supplied tensors reach this layer, SC/initiator.py:125-134 receives typed ciphertext objects instead)."""
        if t is None:
            if optional:
                return None
            raise ValueError(f"{name}: missing array")
        if not isinstance(t, torch.Tensor):
            raise ValueError(f"{name}: expected a torch tensor, got {type(t).__name__}")
        if t.dtype != dtype:
            raise ValueError(f"{name}: dtype {t.dtype}, expected {dtype}")
        if t.device != self.device:
            raise ValueError(f"{name}: lives on {t.device}, the engine works on {self.device}")
        if not t.is_contiguous():
            raise ValueError(f"{name}: not contiguous")
        if t.dim() < 1:
            raise ValueError(f"{name}: scalar tensor")
        if words is not None and t.shape[-1] != words:
            raise ValueError(f"{name}: {t.shape[-1]} words per item, expected {words}")
        if rows is not None:
            have = t.numel() // max(1, t.shape[-1]) if words is not None or t.dim() > 1 else t.numel()
            if not (have == rows or (broadcast and have == 1)):
                raise ValueError(f"{name}: {have} items, expected {rows}")
        return t

    def _result(self, out: torch.Tensor | None, shape: tuple[int, ...], name: str = "out") -> torch.Tensor:
        """The array a step's last launch stores into: a new device array, or the caller's -- on this device, or in PINNED host
        memory (mapped into the device's address space: the kernel's stores cross PCIe as posted writes, and no device-to-host copy
        -- a chip-wide blit kernel on this runtime -- is needed afterwards; wire.reserve hands out such arrays)."""
        if out is None:
            return torch.empty(shape, dtype=torch.int32, device=self.device)
        if not isinstance(out, torch.Tensor) or out.dtype != torch.int32 or tuple(out.shape) != tuple(shape) or not out.is_contiguous():
            raise ValueError(f"{name}: expected a contiguous int32 array of shape {tuple(shape)}")
        if out.device != self.device and not (out.device.type == "cpu" and out.is_pinned()):
            raise ValueError(f"{name}: lives on {out.device}; the engine writes to {self.device} or to pinned host memory")
        return out

    def _host_words(self, x: int, nwords: int):
        arr = int_to_words(x, nwords)
        return arr, arr.ctypes.data_as(C.c_void_p)

    def _host_n_words(self, n: int, nwords: int):
        key = (n, nwords)
        if key not in self._host_n:
            self._host_n[key] = self._host_words(n, nwords)
        return self._host_n[key]

    def upload(self, xs: Iterable[int], nwords: int) -> torch.Tensor:
        arr = ints_to_words(xs, nwords)
        return torch.from_numpy(arr.view(np.int32)).to(self.device)

    def upload_u64(self, xs: Iterable[int]) -> torch.Tensor:
        arr = np.array(list(xs), dtype=np.uint64)
        return torch.from_numpy(arr.view(np.int64)).to(self.device)

    def download(self, t: torch.Tensor) -> list[int]:
        return words_to_ints(t.detach().cpu().numpy().view(np.uint32))

    def upload_words(self, arr: np.ndarray) -> torch.Tensor:
        """A host array of little-endian 32-bit words ([...][nwords], uint32) as a device array of the same shape."""
        arr = np.ascontiguousarray(arr, dtype="<u4")
        if not arr.flags.writeable:                   # (np.frombuffer over bytes: torch refuses to wrap read-only memory quietly)
            arr = arr.copy()
        return torch.from_numpy(arr.view(np.int32)).to(self.device)

    def download_words(self, t: torch.Tensor) -> np.ndarray:
        """A device array as a host array of uint32 words (same shape)."""
        return t.detach().cpu().numpy().view(np.uint32)

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

    def fixed_base_import(self, mod: Modulus, other: "Engine", fb: FixedBase) -> FixedBase:
        """Use a table that `other` (another library context on the same GPU) built for the same modulus: no second copy."""
        if other is self:
            return fb
        fid = C.c_int()
        self._check(self.lib.sc_fbt_import(self.ctx, mod.id, other.ctx, fb.id, C.byref(fid)))
        return FixedBase(fid.value, mod, fb.exp_bits, fb.window)

    def fixed_base_bytes(self, fb: FixedBase) -> int:
        v = C.c_uint64()
        self._check(self.lib.sc_fbt_bytes(self.ctx, fb.id, C.byref(v)))
        return int(v.value)

    # ------------------------------------------------------------------ batched residue arithmetic
    @staticmethod
    def _items(t: torch.Tensor) -> int:
        return t.numel() // max(1, t.shape[-1]) if t.dim() > 1 else 1

    def _out(self, out: torch.Tensor | None, count: int, nwords: int) -> torch.Tensor:
        return self.empty(count, nwords) if out is None else self._arr(out, "out", count, nwords)

    def modmul(self, mod: Modulus, a: torch.Tensor, b: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """a, b: [count][nwords] or a single residue ([nwords] / [1][nwords]) that is broadcast."""
        count = max(self._items(a), self._items(b))
        self._arr(a, "a", count, mod.nwords, broadcast=True)
        self._arr(b, "b", count, mod.nwords, broadcast=True)
        out = self._out(out, count, mod.nwords)
        stride = lambda t: 0 if self._items(t) == 1 and count != 1 else mod.nwords  # noqa: E731
        self._sync_stream()
        self._check(self.lib.sc_modmul(self.ctx, mod.id, self._ptr(a), stride(a), self._ptr(b), stride(b), self._ptr(out), count))
        return out

    def modmul_const(self, mod: Modulus, a: torch.Tensor, c: int, out: torch.Tensor | None = None) -> torch.Tensor:
        count = self._items(a)
        self._arr(a, "a", count, mod.nwords)
        out = self._out(out, count, mod.nwords)
        self._sync_stream()
        self._check(self.lib.sc_modmul_const(self.ctx, mod.id, self._ptr(a), self.constant(mod, c), self._ptr(out), count))
        return out

    def modmul_const_sel(self, mod: Modulus, a: torch.Tensor, c0: int | None, c1: int | None, flags: torch.Tensor,
                         out: torch.Tensor | None = None) -> torch.Tensor:
        """a[i] * (flags[i] ? c1 : c0) mod n; None = the residue 1; flags: uint8 [count]."""
        count = self._items(a)
        self._arr(a, "a", count, mod.nwords)
        self._arr(flags, "flags", dtype=torch.uint8)
        if flags.numel() != count:
            raise ValueError(f"flags: {flags.numel()} items, expected {count}")
        out = self._out(out, count, mod.nwords)
        self._sync_stream()
        k0, k1 = (-1 if c0 is None else self.constant(mod, c0)), (-1 if c1 is None else self.constant(mod, c1))
        self._check(self.lib.sc_modmul_const_sel(self.ctx, mod.id, self._ptr(a), k0, k1, self._ptr(flags), self._ptr(out), count))
        return out

    def modexp_shared(self, mod: Modulus, x: torch.Tensor, e: int, mul_into: torch.Tensor | None = None,
                      out: torch.Tensor | None = None) -> torch.Tensor:
        """x^e mod n for an exponent shared by the batch (key-derived: N, lambda, p - 1, v_p, ...: each distinct exponent is
        registered once and its program kept; per-ciphertext scalars go through modexp_var)."""
        count = self._items(x)
        self._arr(x, "x", count)
        self._arr(mul_into, "mul_into", count, mod.nwords, optional=True)
        out = self._out(out, count, mod.nwords)
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
        count = self._items(x)
        self._arr(x, "x", count)
        self._arr(mul_into, "mul_into", count, mod_m2.nwords, optional=True)
        out = self._out(out, count, mod_m2.nwords)
        self._sync_stream()
        self._check(self.lib.sc_modexp_shared_sq(self.ctx, mod_m.id, mod_m2.id, self.exponent(e), self._ptr(x), x.shape[-1],
                                                 self._ptr(mul_into), self._ptr(out), count))
        return out

    def modexp_shared_isone(self, mod: Modulus, x: torch.Tensor, e: int) -> torch.Tensor:
        count = self._items(x)
        self._arr(x, "x", count)
        flags = torch.empty((count,), dtype=torch.uint8, device=self.device)
        self._sync_stream()
        self._check(self.lib.sc_modexp_shared_isone(self.ctx, mod.id, self.exponent(e), self._ptr(x), x.shape[-1],
                                                    self._ptr(flags), count))
        return flags

    def modexp_shared_isone_any(self, mod: Modulus, x: torch.Tensor, e: int, inner: int) -> torch.Tensor:
        """int64 [inner]: OR over the planes i of (x[i * inner + b]^e == 1) -- x is bit-major [planes * inner][words]."""
        count = self._items(x)
        self._arr(x, "x", count)
        if inner <= 0 or count % inner:
            raise ValueError(f"x: {count} items are not a whole number of planes of {inner}")
        out = torch.empty((inner,), dtype=torch.int64, device=self.device)
        self._sync_stream()
        self._check(self.lib.sc_modexp_shared_isone_any(self.ctx, mod.id, self.exponent(e), self._ptr(x), x.shape[-1], inner,
                                                        self._ptr(out), count))
        return out

    def fixedbase_pow(self, fb: FixedBase, e: torch.Tensor, mul_into: torch.Tensor | None = None,
                      out: torch.Tensor | None = None) -> torch.Tensor:
        count = self._items(e)
        self._arr(e, "e", count)
        self._arr(mul_into, "mul_into", count, fb.mod.nwords, optional=True)
        out = self._out(out, count, fb.mod.nwords)
        self._sync_stream()
        self._check(self.lib.sc_fixedbase_pow(self.ctx, fb.id, self._ptr(e), e.shape[-1], self._ptr(mul_into),
                                              self._ptr(out), count))
        return out

    def modexp_var(self, mod: Modulus, x: torch.Tensor, e: torch.Tensor, ebits: int, fb: FixedBase | None = None,
                   e2: torch.Tensor | None = None, out: torch.Tensor | None = None, dest: torch.Tensor | None = None) -> torch.Tensor:
        """x[i]^e[i] [* base^e2[i]]; with `dest` (int64 [count]) the result of item i lands in row dest[i] (a permutation)."""
        count = self._items(x)
        self._arr(x, "x", count, mod.nwords)
        self._arr(e, "e", count)
        self._arr(e2, "e2", count, optional=fb is None)
        out = self._out(out, count, mod.nwords)
        self._sync_stream()
        fbid, e2w = (-1 if fb is None else fb.id), (0 if e2 is None else e2.shape[-1])
        if dest is None:
            self._check(self.lib.sc_modexp_var(self.ctx, mod.id, self._ptr(x), self._ptr(e), e.shape[-1], ebits, fbid, self._ptr(e2), e2w,
                                               self._ptr(out), count))
        else:
            self._arr(dest, "dest", dtype=torch.int64)
            if dest.numel() != count:
                raise ValueError(f"dest: {dest.numel()} items, expected {count}")
            self._check(self.lib.sc_modexp_var_scatter(self.ctx, mod.id, self._ptr(x), self._ptr(e), e.shape[-1], ebits, fbid, self._ptr(e2), e2w,
                                                       self._ptr(dest), self._ptr(out), count))
        return out

    def modinv(self, mod: Modulus, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        count = self._items(x)
        self._arr(x, "x", count, mod.nwords)
        out = self._out(out, count, mod.nwords)
        self._sync_stream()
        bad = C.c_int64(-1)
        rc = self.lib.sc_modinv(self.ctx, mod.id, self._ptr(x), self._ptr(out), count, C.byref(bad))
        if rc == -3:
            raise NotInvertibleError(self.lib.sc_last_error(self.ctx).decode(), int(bad.value))
        self._check(rc)
        return out

    def paillier_encrypt_raw(self, mod_n2: Modulus, n: int, m: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        count = self._items(m)
        self._arr(m, "m", count)
        out = self._out(out, count, mod_n2.nwords)
        self._sync_stream()
        self._check(self.lib.sc_paillier_encrypt_raw(self.ctx, mod_n2.id, self.constant(mod_n2, n), self._ptr(m),
                                                     m.shape[-1], self._ptr(out), count))
        return out

    def paillier_encrypt_raw_neg(self, mod_n2: Modulus, n: int, m: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """[[-m]] = 1 - m N mod N^2 (the inverse of paillier_encrypt_raw's result, without an inversion)."""
        count = self._items(m)
        self._arr(m, "m", count)
        out = self._out(out, count, mod_n2.nwords)
        self._sync_stream()
        self._check(self.lib.sc_paillier_encrypt_raw_neg(self.ctx, mod_n2.id, self.constant(mod_n2, n), self._ptr(m),
                                                         m.shape[-1], self._ptr(out), count))
        return out

    def paillier_l_mul(self, mod: Modulus, k: int, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        count = self._items(x)
        self._arr(x, "x", count)
        out = self._out(out, count, mod.nwords)
        self._sync_stream()
        self._check(self.lib.sc_paillier_l_mul(self.ctx, mod.id, self.constant(mod, k), self._ptr(x), x.shape[-1],
                                               self._ptr(out), count))
        return out

    def crt_combine(self, mod_p: Modulus, mod_full: Modulus, mq: int, a_p: torch.Tensor, a_q: torch.Tensor,
                    out: torch.Tensor | None = None) -> torch.Tensor:
        """x with x = a_p (mod m_p), x = a_q (mod m_q), m_p m_q = mod_full.n (CRT recombination on the GPU)."""
        count = self._items(a_p)
        self._arr(a_p, "a_p", count)
        self._arr(a_q, "a_q", count)
        out = self._out(out, count, mod_full.nwords)
        key = (mq, mod_p.n)
        if key not in self._crt_k:
            self._crt_k[key] = pow(mq, -1, mod_p.n)
        k = self._crt_k[key]
        self._sync_stream()
        self._check(self.lib.sc_crt_combine(self.ctx, mod_p.id, mod_full.id, self.constant(mod_p, k), self.constant(mod_p, mod_p.n - k),
                                            self.constant(mod_full, mq), self._ptr(a_p), a_p.shape[-1], self._ptr(a_q), a_q.shape[-1],
                                            self._ptr(out), count))
        return out

    def plain_alice(self, r: torch.Tensor, n: int, l: int):
        self._arr(r, "r")
        if r.dim() != 2:
            raise ValueError("r: expected [count][nwords]")
        count, nw = r.shape
        m1 = self.empty(count, nw + 1)
        alpha = torch.empty((count,), dtype=torch.int64, device=self.device)
        alpha_t = torch.empty_like(alpha)
        rsmall = torch.empty_like(alpha)
        rshift = self.empty(count, nw)
        arr, p = self._host_n_words(n, nw)
        self._sync_stream()
        self._check(self.lib.sc_plain_alice(self.ctx, self._ptr(r), p, nw, l, count, self._ptr(m1), self._ptr(alpha),
                                            self._ptr(alpha_t), self._ptr(rsmall), self._ptr(rshift)))
        return m1, alpha, alpha_t, rsmall, rshift

    def plain_bob(self, z: torch.Tensor, n: int, l: int):
        self._arr(z, "z")
        if z.dim() != 2:
            raise ValueError("z: expected [count][nwords]")
        count, nw = z.shape
        beta = torch.empty((count,), dtype=torch.int64, device=self.device)
        dbit = torch.empty_like(beta)
        zeta1 = self.empty(count, nw)
        zeta2 = self.empty(count, nw)
        arr, p = self._host_n_words(n, nw)
        self._sync_stream()
        self._check(self.lib.sc_plain_bob(self.ctx, self._ptr(z), p, nw, l, count, self._ptr(beta), self._ptr(dbit),
                                          self._ptr(zeta1), self._ptr(zeta2)))
        return beta, dbit, zeta1, zeta2

    def dgk_step4(self, mod: Modulus, g: int, g_inv: int, l: int, beta: torch.Tensor, beta_inv: torch.Tensor,
                  d: torch.Tensor, d_inv: torch.Tensor, alpha: torch.Tensor, alpha_tilde: torch.Tensor,
                  rsmall: torch.Tensor, delta_a: torch.Tensor) -> torch.Tensor:
        count = self._items(d)
        nw = mod.nwords
        self._arr(d, "d", count, nw)
        self._arr(d_inv, "d_inv", count, nw)
        self._arr(beta, "beta", l * count, nw)
        self._arr(beta_inv, "beta_inv", l * count, nw)
        for name, t in (("alpha", alpha), ("alpha_tilde", alpha_tilde), ("rsmall", rsmall), ("delta_a", delta_a)):
            self._arr(t, name, dtype=torch.int64)
            if t.numel() != count:
                raise ValueError(f"{name}: {t.numel()} items, expected {count}")
        out = torch.empty((l + 1, count, nw), dtype=torch.int32, device=self.device)
        self._sync_stream()
        self._check(self.lib.sc_dgk_step4(self.ctx, mod.id, self.constant(mod, g), self.constant(mod, g_inv), l,
                                          self._ptr(beta), self._ptr(beta_inv), self._ptr(d), self._ptr(d_inv),
                                          self._ptr(alpha), self._ptr(alpha_tilde), self._ptr(rsmall), self._ptr(delta_a),
                                          self._ptr(out), count))
        return out

    # ------------------------------------------------------------------ multi-GPU through the C ABI (one engine per GPU / process)
    def comm_unique_id(self) -> bytes:
        """Rendezvous id for sc_comm_init (rank 0 creates it and ships it to the other ranks)."""
        buf = (C.c_char * 128)()
        self._check(self.lib.sc_comm_unique_id(self.ctx, C.cast(buf, C.c_void_p)))
        return bytes(buf)

    def comm_init(self, comm_id: bytes, rank: int, nranks: int) -> None:
        if len(comm_id) != 128:
            raise ValueError("the rendezvous id has 128 bytes")
        buf = (C.c_char * 128).from_buffer_copy(comm_id)
        self._check(self.lib.sc_comm_init(self.ctx, C.cast(buf, C.c_void_p), int(rank), int(nranks)))
        self._comm_nranks = int(nranks)

    def allgather(self, local: torch.Tensor) -> torch.Tensor:
        """All ranks' `local` arrays (same shape, int32 words) stacked in rank order along the first axis: one RCCL all-gather on
        the engine's stream (sc_allgather)."""
        n = getattr(self, "_comm_nranks", 0)
        if n < 1:
            raise ScError("no communicator: call comm_init first")
        self._arr(local, "local")
        out = torch.empty((n * local.shape[0],) + tuple(local.shape[1:]), dtype=torch.int32, device=self.device)
        self._sync_stream()
        self._check(self.lib.sc_allgather(self.ctx, self._ptr(local), self._ptr(out), local.numel()))
        return out

    def comm_destroy(self) -> None:
        self._check(self.lib.sc_comm_destroy(self.ctx))
        self._comm_nranks = 0

    # ------------------------------------------------------------------ scheme-level entry points (one library call each)
    def _words_arg(self, x: int | None, nwords: int):
        if x is None:
            return None, C.c_void_p(0)
        return self._host_words(x, nwords)

    def _mod_handle(self, mid: int, n: int, nwords: int) -> Modulus:
        return self._mods.setdefault((n, nwords), Modulus(mid, n, nwords))

    def paillier_key(self, n: int, p: int | None = None, q: int | None = None, use_crt: bool = True, use_pairs: bool = True) -> PaillierKey:
        """Library-side key object: every modulus / exponent / constant the scheme derives, the key holder's CRT included."""
        ck = (n, p, q, bool(use_crt), bool(use_pairs))
        if ck in self._pkeys:
            return self._pkeys[ck]
        nw = (n.bit_length() + 31) // 32
        pw = 0 if p is None else (max(p.bit_length(), q.bit_length()) + 31) // 32
        a_n, p_n = self._host_words(n, nw)
        a_p, p_p = self._words_arg(p, pw)
        a_q, p_q = self._words_arg(q, pw)
        kid, m1, m2 = C.c_int(), C.c_int(), C.c_int()
        self._sync_stream()
        self._check(self.lib.sc_paillier_key_create(self.ctx, p_n, nw, p_p, p_q, pw, (0 if use_crt else 1) | (0 if use_pairs else 2), C.byref(kid)))
        self._check(self.lib.sc_paillier_key_mods(self.ctx, kid.value, C.byref(m1), C.byref(m2)))
        self._pkeys[ck] = PaillierKey(kid.value, self._mod_handle(m1.value, n, nw), self._mod_handle(m2.value, n * n, 2 * nw), p is not None)
        return self._pkeys[ck]

    def paillier_encrypt(self, key: PaillierKey, m: torch.Tensor, negate: bool = False, out: torch.Tensor | None = None) -> torch.Tensor:
        count = self._items(m)
        self._arr(m, "m", count)
        out = self._out(out, count, key.mod_n2.nwords)
        self._sync_stream()
        self._check(self.lib.sc_paillier_encrypt(self.ctx, key.id, self._ptr(m), m.shape[-1], int(negate), self._ptr(out), count))
        return out

    def paillier_randomize(self, key: PaillierKey, c: torch.Tensor | None, rho: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """c * rho^N mod N^2 (c None: the randomizers): ct.randomize() for a batch."""
        count = self._items(rho)
        self._arr(rho, "rho", count, key.mod_n.nwords)
        self._arr(c, "c", count, key.mod_n2.nwords, optional=True)
        out = self._out(out, count, key.mod_n2.nwords)
        self._sync_stream()
        self._check(self.lib.sc_paillier_randomize(self.ctx, key.id, self._ptr(c), self._ptr(rho), self._ptr(out), count))
        return out

    def paillier_decrypt(self, key: PaillierKey, c: torch.Tensor) -> torch.Tensor:
        count = self._items(c)
        self._arr(c, "c", count, key.mod_n2.nwords)
        out = self.empty(count, key.mod_n.nwords)
        self._sync_stream()
        self._check(self.lib.sc_paillier_decrypt(self.ctx, key.id, self._ptr(c), self._ptr(out), count))
        return out

    def dgk_key(self, n: int, g: int, h: int, u: int, t: int, p: int | None = None, q: int | None = None, v_p: int | None = None,
                v_q: int | None = None, randomizer_bits: int = 400, window: int = 8, use_crt: bool = True,
                table_source: "tuple[Engine, DgkKey] | None" = None) -> DgkKey:
        """Library-side DGK key object; builds the fixed-base tables for h (or takes `table_source`'s over, read-only)."""
        ck = (n, g, h, u, t, p, q, v_p, v_q, int(randomizer_bits), int(window), bool(use_crt))
        if ck in self._dkeys:          # the same key, tables included, whoever asks (and wherever its tables came from)
            return self._dkeys[ck]
        nw, uw = (n.bit_length() + 31) // 32, (u.bit_length() + 31) // 32
        pw = 0 if p is None else (max(p.bit_length(), q.bit_length()) + 31) // 32
        vw = 0 if v_p is None else (max(v_p.bit_length(), v_q.bit_length()) + 31) // 32
        keep = [self._host_words(x, nw) for x in (n, g % n, h % n)] + [self._host_words(u, uw)]
        sec = [self._words_arg(x, w) for x, w in ((p, pw), (q, pw), (v_p, vw), (v_q, vw))]
        if table_source is not None and table_source[0] is self:
            self._dkeys[ck] = table_source[1]
            return table_source[1]
        src_ctx, src_key = (C.c_void_p(0), -1) if table_source is None else (table_source[0].ctx, table_source[1].id)
        kid, m_n, m_p = C.c_int(), C.c_int(), C.c_int()
        self._sync_stream()
        self._check(self.lib.sc_dgk_key_create(self.ctx, keep[0][1], keep[1][1], keep[2][1], nw, keep[3][1], uw, int(t), sec[0][1], sec[1][1], pw,
                                               sec[2][1], sec[3][1], vw, int(randomizer_bits), int(window), 0 if use_crt else 1, src_ctx, src_key,
                                               C.byref(kid)))
        self._check(self.lib.sc_dgk_key_info(self.ctx, kid.value, C.byref(m_n), C.byref(m_p), None))
        mod_p = None if p is None else self._mod_handle(m_p.value, p, (p.bit_length() + 31) // 32)
        self._dkeys[ck] = DgkKey(kid.value, self._mod_handle(m_n.value, n, nw), mod_p, u, int(randomizer_bits), int(window), p is not None)
        return self._dkeys[ck]

    def dgk_table_bytes(self, key: DgkKey) -> int:
        v = C.c_uint64()
        self._check(self.lib.sc_dgk_key_info(self.ctx, key.id, None, None, C.byref(v)))
        return int(v.value)

    def dgk_randomize(self, key: DgkKey, c: torch.Tensor | None, r: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """c * h^r mod n (c None: the randomizers)."""
        count = self._items(r)
        self._arr(r, "r", count)
        self._arr(c, "c", count, key.mod_n.nwords, optional=True)
        out = self._out(out, count, key.mod_n.nwords)
        self._sync_stream()
        self._check(self.lib.sc_dgk_randomize(self.ctx, key.id, self._ptr(c), self._ptr(r), r.shape[-1], self._ptr(out), count))
        return out

    def dgk_encrypt_bits_randomized(self, key: DgkKey, bits: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
        """g^bits[i] * h^r[i] mod n; bits: uint8 [count]."""
        count = self._items(r)
        self._arr(r, "r", count)
        self._arr(bits, "bits", dtype=torch.uint8)
        if bits.numel() != count:
            raise ValueError(f"bits: {bits.numel()} items, expected {count}")
        out = self.empty(count, key.mod_n.nwords)
        self._sync_stream()
        self._check(self.lib.sc_dgk_encrypt_bits_randomized(self.ctx, key.id, self._ptr(bits), self._ptr(r), r.shape[-1], self._ptr(out), count))
        return out

    def dgk_is_zero(self, key: DgkKey, c: torch.Tensor) -> torch.Tensor:
        count = self._items(c)
        self._arr(c, "c", count, key.mod_n.nwords)
        flags = torch.empty((count,), dtype=torch.uint8, device=self.device)
        self._sync_stream()
        self._check(self.lib.sc_dgk_is_zero(self.ctx, key.id, self._ptr(c), self._ptr(flags), count))
        return flags

    def dgk_any_zero(self, key: DgkKey, c: torch.Tensor) -> torch.Tensor:
        """int64 [B]: 1 where some plane of c [planes][B][nw] decrypts to zero (KeyHolder.step_4j)."""
        self._arr(c, "c", words=key.mod_n.nwords)
        if c.dim() != 3:
            raise ValueError("c: expected [planes][B][nwords]")
        planes, inner, _ = c.shape
        out = torch.empty((inner,), dtype=torch.int64, device=self.device)
        self._sync_stream()
        self._check(self.lib.sc_dgk_any_zero(self.ctx, key.id, self._ptr(c), planes, inner, self._ptr(out)))
        return out

    def _flags(self, count: int, **arrays: torch.Tensor) -> None:
        for name, t in arrays.items():
            self._arr(t, name, dtype=torch.int64)
            if t.numel() != count:
                raise ValueError(f"{name}: {t.numel()} items, expected {count}")

    def initiator_step1(self, key: PaillierKey, l: int, x_enc: torch.Tensor, y_enc: torch.Tensor, r: torch.Tensor,
                        rho_z: torch.Tensor | None = None, ready: bool = False, out: torch.Tensor | None = None):
        """(z_enc, alpha, alpha_tilde, r_small, r_shift): Initiator.step_1 / step_3 for a batch, [[z]] randomized with rho_z^N
        (`ready`: rho_z holds the finished randomizers rho_z^N mod N^2, [B][2nw], computed ahead of time)."""
        count = self._items(x_enc)
        nw = key.mod_n.nwords
        self._arr(x_enc, "x_enc", count, 2 * nw)
        self._arr(y_enc, "y_enc", count, 2 * nw)
        self._arr(r, "r", count, nw)
        self._arr(rho_z, "rho_z", count, 2 * nw if ready else nw, optional=True)
        z = self._result(out, (count, 2 * nw), "z_out")
        alpha = torch.empty((count,), dtype=torch.int64, device=self.device)
        alpha_t, rsmall = torch.empty_like(alpha), torch.empty_like(alpha)
        rshift = self.empty(count, nw)
        self._sync_stream()
        rc = self.lib.sc_initiator_step1(self.ctx, key.id, int(l), self._ptr(x_enc), self._ptr(y_enc), self._ptr(r), self._ptr(rho_z), int(ready),
                                         self._ptr(z), self._ptr(alpha), self._ptr(alpha_t), self._ptr(rsmall), self._ptr(rshift), count)
        self._check(rc)
        return z, alpha, alpha_t, rsmall, rshift

    def keyholder_step2_4b(self, pkey: PaillierKey, dkey: DgkKey, l: int, z_enc: torch.Tensor, r_rand: torch.Tensor | None = None,
                           ready: bool = False, out: torch.Tensor | None = None):
        """(z, beta, d, zeta_1, zeta_2, [d],[beta_i] as [l+1][B][nw]): KeyHolder.step_2 / 4a / 4b (+ their randomizations)."""
        count = self._items(z_enc)
        nw, nd = pkey.mod_n.nwords, dkey.mod_n.nwords
        self._arr(z_enc, "z_enc", count, 2 * nw)
        self._arr(r_rand, "r_rand", (l + 1) * count, nd if ready else None, optional=True)
        z, zeta1, zeta2 = self.empty(count, nw), self.empty(count, nw), self.empty(count, nw)
        beta = torch.empty((count,), dtype=torch.int64, device=self.device)
        dbit = torch.empty_like(beta)
        out = self._result(out, (l + 1, count, nd), "d_beta_out")
        self._sync_stream()
        self._check(self.lib.sc_keyholder_step2_4b(self.ctx, pkey.id, dkey.id, int(l), self._ptr(z_enc), self._ptr(r_rand),
                                                   0 if r_rand is None else r_rand.shape[-1], int(ready), self._ptr(z), self._ptr(beta), self._ptr(dbit),
                                                   self._ptr(zeta1), self._ptr(zeta2), self._ptr(out), count))
        return z, beta, dbit, zeta1, zeta2, out

    def initiator_step4(self, key: DgkKey, l: int, d_enc: torch.Tensor, beta_enc: torch.Tensor, alpha: torch.Tensor, alpha_tilde: torch.Tensor,
                        rsmall: torch.Tensor, delta_a: torch.Tensor, rhos: torch.Tensor | None = None, permutation: torch.Tensor | None = None,
                        r_rand: torch.Tensor | None = None, want_unblinded: bool = False, ready: bool = False, out: torch.Tensor | None = None):
        """(c, c after step 4h or None): Initiator.step_4c .. 4i for a batch; see sc_initiator_step4 (`ready`: r_rand holds h^r)."""
        count = self._items(d_enc)
        nw = key.mod_n.nwords
        self._arr(d_enc, "d_enc", count, nw)
        self._arr(beta_enc, "beta_enc", l * count, nw)
        self._flags(count, alpha=alpha, alpha_tilde=alpha_tilde, rsmall=rsmall, delta_a=delta_a)
        self._arr(rhos, "rhos", (l + 1) * count, optional=True)
        self._arr(r_rand, "r_rand", (l + 1) * count, nw if ready else None, optional=True)
        if permutation is not None:
            self._arr(permutation, "permutation", dtype=torch.int64)
            if tuple(permutation.shape) != (count, l + 1):
                raise ValueError(f"permutation: expected int64 [{count}][{l + 1}], got {tuple(permutation.shape)}")
        out = self._result(out, (l + 1, count, nw), "c_out")
        mid = torch.empty((l + 1, count, nw), dtype=torch.int32, device=self.device) if (want_unblinded and rhos is not None) else None
        self._sync_stream()
        rc = self.lib.sc_initiator_step4(self.ctx, key.id, int(l), self._ptr(d_enc), self._ptr(beta_enc), self._ptr(alpha), self._ptr(alpha_tilde),
                                         self._ptr(rsmall), self._ptr(delta_a), self._ptr(rhos), 0 if rhos is None else rhos.shape[-1],
                                         self._ptr(permutation), self._ptr(r_rand), 0 if r_rand is None else r_rand.shape[-1],
                                         int(ready), self._ptr(mid), self._ptr(out), count)
        self._check(rc)
        return out, mid

    def initiator_step4i(self, key: DgkKey, l: int, c_in: torch.Tensor, rhos: torch.Tensor, permutation: torch.Tensor | None = None,
                         r_rand: torch.Tensor | None = None, ready: bool = False) -> torch.Tensor:
        """Blinding c_i^rho_i [* h^r_i] and the per-comparison shuffle of a vector [l+1][B][nw] that is already there."""
        nw = key.mod_n.nwords
        if c_in.dim() != 3 or c_in.shape[0] != l + 1:
            raise ValueError(f"c: expected [{l + 1}][B][{nw}], got {tuple(c_in.shape)}")
        count = c_in.shape[1]
        self._arr(c_in, "c", (l + 1) * count, nw)
        self._arr(rhos, "rhos", (l + 1) * count)
        self._arr(r_rand, "r_rand", (l + 1) * count, nw if ready else None, optional=True)
        if permutation is not None:
            self._arr(permutation, "permutation", dtype=torch.int64)
            if tuple(permutation.shape) != (count, l + 1):
                raise ValueError(f"permutation: expected int64 [{count}][{l + 1}], got {permutation.dtype} {tuple(permutation.shape)}")
        out = torch.empty_like(c_in)
        self._sync_stream()
        self._check(self.lib.sc_initiator_step4i(self.ctx, key.id, int(l), self._ptr(c_in), self._ptr(rhos), rhos.shape[-1], self._ptr(permutation),
                                                 self._ptr(r_rand), 0 if r_rand is None else r_rand.shape[-1], int(ready), self._ptr(out), count))
        return out

    def keyholder_step4j_5(self, pkey: PaillierKey, dkey: DgkKey, l: int, c_enc: torch.Tensor, zeta1: torch.Tensor, zeta2: torch.Tensor,
                           rho3: torch.Tensor | None = None, ready: bool = False, out: torch.Tensor | None = None):
        """(delta_B int64 [B], [[zeta_1]] | [[zeta_2]] | [[delta_B]] as [3B][2nw]): KeyHolder.step_4j / step_5 (+ randomizations)."""
        nw, nd = pkey.mod_n.nwords, dkey.mod_n.nwords
        count = self._items(zeta1)
        self._arr(c_enc, "c_enc", (l + 1) * count, nd)
        self._arr(zeta1, "zeta_1", count, nw)
        self._arr(zeta2, "zeta_2", count, nw)
        self._arr(rho3, "rho3", 3 * count, 2 * nw if ready else nw, optional=True)
        delta_b = torch.empty((count,), dtype=torch.int64, device=self.device)
        out = self._result(out, (3 * count, 2 * nw), "out3")
        self._sync_stream()
        self._check(self.lib.sc_keyholder_step4j_5(self.ctx, pkey.id, dkey.id, int(l), self._ptr(c_enc), self._ptr(zeta1), self._ptr(zeta2),
                                                   self._ptr(rho3), int(ready), self._ptr(delta_b), self._ptr(out), count))
        return delta_b, out

    def initiator_step67(self, key: PaillierKey, delta_a: torch.Tensor, delta_b_enc: torch.Tensor, zeta1_enc: torch.Tensor, zeta2_enc: torch.Tensor,
                         rsmall: torch.Tensor, rshift: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        count = self._items(delta_b_enc)
        nw = key.mod_n.nwords
        for name, t in (("delta_b_enc", delta_b_enc), ("zeta_1_enc", zeta1_enc), ("zeta_2_enc", zeta2_enc)):
            self._arr(t, name, count, 2 * nw)
        self._arr(rshift, "r_shift", count, nw)
        self._flags(count, delta_a=delta_a, rsmall=rsmall)
        out = self._out(out, count, 2 * nw)
        self._sync_stream()
        rc = self.lib.sc_initiator_step67(self.ctx, key.id, self._ptr(delta_a), self._ptr(delta_b_enc), self._ptr(zeta1_enc), self._ptr(zeta2_enc),
                                          self._ptr(rsmall), self._ptr(rshift), 0, self._ptr(out), count)
        self._check(rc)
        return out

    # ------------------------------------------------------------------ device-side CSPRNG (sc_rng_*)
    def rng_seed(self, key: bytes | None = None) -> None:
        """Key the context's generator: 32 bytes from the caller (reproducible tests) or, with None, from the OS.  An unseeded
        engine seeds itself from the OS on its first draw."""
        if key is not None and len(key) != 32:
            raise ValueError("the generator key has 32 bytes")
        buf = None if key is None else (C.c_char * 32).from_buffer_copy(key)
        self._check(self.lib.sc_rng_seed(self.ctx, None if buf is None else C.cast(buf, C.c_void_p)))

    def rng_bits(self, bits: int, count: int) -> torch.Tensor:
        """[count][ceil(bits/32)] words, each item uniform below 2^bits."""
        out = self.empty(count, (bits + 31) // 32)
        self._sync_stream()
        self._check(self.lib.sc_rng_bits(self.ctx, int(bits), self._ptr(out), count))
        return out

    def rng_below(self, n: int, count: int, nonzero: bool = False) -> torch.Tensor:
        """[count][nwords(n)] words, each item uniform in [0, n) (or [1, n)): rejection sampling on the device."""
        nw = (n.bit_length() + 31) // 32
        arr, p = self._host_n_words(n, nw)
        out = self.empty(count, nw)
        self._sync_stream()
        self._check(self.lib.sc_rng_below(self.ctx, p, nw, int(bool(nonzero)), self._ptr(out), count))
        return out

    def rng_coins(self, count: int) -> torch.Tensor:
        """int64 [count], each 0 or 1."""
        out = torch.empty((count,), dtype=torch.int64, device=self.device)
        self._sync_stream()
        self._check(self.lib.sc_rng_coins(self.ctx, self._ptr(out), count))
        return out

    def rng_permutations(self, k: int, count: int) -> torch.Tensor:
        """int64 [count][k]: one uniform permutation of range(k) per item."""
        out = torch.empty((count, k), dtype=torch.int64, device=self.device)
        self._sync_stream()
        self._check(self.lib.sc_rng_permutations(self.ctx, int(k), self._ptr(out), count))
        return out

    def peak_probe(self) -> float:
        v = C.c_double()
        self._sync_stream()
        self._check(self.lib.sc_peak_probe(self.ctx, C.byref(v)))
        return v.value

    def policy(self) -> dict:
        """The measured constants of the one-lane policy on this device (sc_ctx_policy; measured once per device and process)."""
        v = (C.c_double * 6)()
        self._sync_stream()
        self._check(self.lib.sc_ctx_policy(self.ctx, v))
        names = ("one_lane_full_round_ms", "one_lane_half_round_ms", "two_lane_full_round_ms", "two_lane_later_half_round_ms", "two_lane_only_half_round_ms", "simds")
        return dict(zip(names, list(v)))

    def clock_probe(self, key: "PaillierKey", rho: torch.Tensor) -> tuple[float, float]:
        """(engine clock in GHz held by the dominant pair launch, its duration in ms) from the stamping twin of k_pvm<4,18,neg1>
        (sc_clock_probe): rho^N mod N^2 for rho [count][nwords] -- a diagnostic, never part of a timed region."""
        count = self._items(rho)
        self._arr(rho, "rho", count, key.mod_n.nwords)
        ghz, ms = C.c_double(), C.c_double()
        self._sync_stream()
        self._check(self.lib.sc_clock_probe(self.ctx, key.id, self._ptr(rho), count, C.byref(ghz), C.byref(ms)))
        return ghz.value, ms.value

    def set_latency_mode(self, mode: int) -> None:
        """Small-batch kernel policy (sc_ctx_set_latency_mode): 0 never, 1 automatic (default), 2 whenever available."""
        self._check(self.lib.sc_ctx_set_latency_mode(self.ctx, int(mode)))

    def set_onelane_mode(self, mode: int) -> None:
        """Large-batch kernel policy for moduli up to 1028 bits (sc_ctx_set_onelane_mode): 0 never, 1 automatic, 2 whenever it fits."""
        self._check(self.lib.sc_ctx_set_onelane_mode(self.ctx, int(mode)))

    def set_fork_mode(self, mode: int) -> None:
        """Fork / join of the CRT's q-side inside one call on small batches (sc_ctx_set_fork_mode): 0 never, 1 automatic (default)."""
        self._check(self.lib.sc_ctx_set_fork_mode(self.ctx, int(mode)))

    def set_pair_policy(self, hold_ms: float = 5.0, max_rounds: float = 2.5) -> None:
        """Segments of long pair launches on a shared chip (sc_ctx_set_pair_policy): a resident wave holds its slot for about `hold_ms`
        (0: never cut); launches of more than `max_rounds` rounds stay whole."""
        self._check(self.lib.sc_ctx_set_pair_policy(self.ctx, float(hold_ms), float(max_rounds)))

    def stats(self) -> dict:
        """Counters of the context (sc_ctx_stats)."""
        v = (C.c_uint64 * 3)()
        self._check(self.lib.sc_ctx_stats(self.ctx, v, 3))
        return {"segmented_pair_launches": int(v[0]), "pair_segments": int(v[1]), "pair_calibrations": int(v[2])}

    def set_chip_share(self, contexts: int) -> None:
        """This engine shares its GPU with contexts - 1 other engines working at the same time (sc_ctx_set_chip_share)."""
        self._check(self.lib.sc_ctx_set_chip_share(self.ctx, int(contexts)))

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
