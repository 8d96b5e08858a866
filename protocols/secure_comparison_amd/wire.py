"""Compact binary wire format for ciphertext batches.

The reference sends Python scheme / ciphertext objects through its transport's serializer, one object per
ciphertext (SC/test/conftest.py:182,198).  A batch of B comparisons moves ~19 KB per comparison, so batches travel
as raw little-endian word arrays with a 16+ byte header instead:

    magic "SCB1" | dtype code u8 | ndim u8 | reserved u16 | dims (ndim x u64) | payload

Public keys travel as a small JSON document (integers as hex)."""
from __future__ import annotations

import json
import struct

import numpy as np
import torch

MAGIC = b"SCB1"
_DTYPES = {0: (torch.int32, np.dtype("<i4")), 1: (torch.int64, np.dtype("<i8")), 2: (torch.uint8, np.dtype("u1"))}
_CODES = {t: c for c, (t, _) in _DTYPES.items()}


def pack_tensor(t: torch.Tensor) -> bytes:
    if t.dtype not in _CODES:
        raise ValueError(f"unsupported dtype {t.dtype}")
    arr = t.detach().contiguous().cpu().numpy()
    head = MAGIC + struct.pack("<BBH", _CODES[t.dtype], arr.ndim, 0) + struct.pack(f"<{arr.ndim}Q", *arr.shape)
    return head + arr.astype(_DTYPES[_CODES[t.dtype]][1], copy=False).tobytes()


MAX_NDIM = 4


def unpack_tensor(buf: bytes, device: torch.device | str = "cpu") -> torch.Tensor:
    """Parse one array message.  The header comes from the peer: everything in it is checked against the bytes that
    actually arrived before an array of that shape is built (ValueError otherwise)."""
    if len(buf) < 8 or buf[:4] != MAGIC:
        raise ValueError("not a secure-comparison batch message")
    code, ndim, _ = struct.unpack_from("<BBH", buf, 4)
    if code not in _DTYPES:
        raise ValueError(f"unknown dtype code {code} in batch message")
    if ndim > MAX_NDIM or len(buf) < 8 + 8 * ndim:
        raise ValueError("malformed batch message header")
    shape = struct.unpack_from(f"<{ndim}Q", buf, 8)
    tdt, ndt = _DTYPES[code]
    count = 1
    for d in shape:
        count *= d
    if count * ndt.itemsize != len(buf) - 8 - 8 * ndim:
        raise ValueError(f"batch message announces shape {tuple(shape)} but carries {len(buf) - 8 - 8 * ndim} payload bytes")
    arr = np.frombuffer(buf, dtype=ndt, offset=8 + 8 * ndim).reshape(shape)
    return torch.from_numpy(arr.copy()).to(device)


def expect_array(t: torch.Tensor, shape: tuple[int, ...], name: str, dtype: torch.dtype = torch.int32) -> torch.Tensor:
    """A received array must have exactly the dtype and shape this party's own parameters (l, B, key sizes) dictate: the
    kernels take sizes from the local side, never from the message."""
    if t.dtype != dtype:
        raise ValueError(f"{name}: received dtype {t.dtype}, expected {dtype}")
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: received shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t.contiguous()


def pack_many(*tensors: torch.Tensor) -> bytes:
    parts = [pack_tensor(t) for t in tensors]
    return struct.pack("<I", len(parts)) + b"".join(struct.pack("<Q", len(p)) + p for p in parts)


def unpack_many(buf: bytes, device: torch.device | str = "cpu", expect: int | None = None) -> list[torch.Tensor]:
    if len(buf) < 4:
        raise ValueError("malformed batch message")
    (n,) = struct.unpack_from("<I", buf, 0)
    if expect is not None and n != expect:
        raise ValueError(f"batch message carries {n} arrays, expected {expect}")
    if n > 64:
        raise ValueError("malformed batch message (array count)")
    off, out = 4, []
    for _ in range(n):
        if off + 8 > len(buf):
            raise ValueError("truncated batch message")
        (ln,) = struct.unpack_from("<Q", buf, off)
        off += 8
        if off + ln > len(buf):
            raise ValueError("truncated batch message")
        out.append(unpack_tensor(buf[off:off + ln], device))
        off += ln
    return out


def pack_public_schemes(paillier, dgk) -> bytes:
    pk = dgk.public_key
    doc = {"paillier": {"n": hex(paillier.public_key.n)},
           "dgk": {"n": hex(pk.n), "g": hex(pk.g), "h": hex(pk.h), "u": hex(pk.u), "t": pk.t,
                   "randomizer_bits": dgk.randomizer_bits, "fixed_base_window": dgk.fixed_base_window}}
    return json.dumps(doc).encode()


def unpack_public_schemes(buf: bytes, engine=None):
    from .schemes import DGK, Paillier

    doc = json.loads(buf.decode())
    d = doc["dgk"]
    return (Paillier(int(doc["paillier"]["n"], 16), engine=engine),
            DGK(int(d["n"], 16), int(d["g"], 16), int(d["h"], 16), int(d["u"], 16), d["t"], engine=engine,
                randomizer_bits=d["randomizer_bits"], fixed_base_window=d["fixed_base_window"]))
