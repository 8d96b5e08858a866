"""Host-side generation of the random inputs of a batch.

`source="os"` draws from the operating system's CSPRNG (the reference uses `secrets`, SC/initiator.py:5);
`source="torch"` uses a seeded torch generator (on its own device) -- NOT cryptographic, for benchmarks and reproducible
tests only; every helper here honours it, so boot_pools(source="torch", generator=g) is reproducible."""
from __future__ import annotations

import os

import numpy as np
import torch


def _os_words(shape: tuple[int, ...]) -> np.ndarray:
    n = int(np.prod(shape))
    return np.frombuffer(os.urandom(4 * n), dtype="<u4").reshape(shape).copy()


def random_bits(bits: int, lead_shape: tuple[int, ...], device, source: str = "os", generator: torch.Generator | None = None) -> torch.Tensor:
    """Uniform integers below 2^bits as word arrays [*lead_shape][ceil(bits/32)] (int32 view of uint32)."""
    nw = (bits + 31) // 32
    top = bits - 32 * (nw - 1)
    if source == "os":
        w = _os_words(lead_shape + (nw,))
        if top < 32:
            w[..., -1] &= (1 << top) - 1
        return torch.from_numpy(w.view(np.int32)).to(device)
    w = torch.randint(-(2 ** 31), 2 ** 31, lead_shape + (nw,), generator=generator, device=device, dtype=torch.int64).to(torch.int32)
    if top < 32:
        w[..., -1] &= (1 << top) - 1
    return w


def uniform_below(n: int, count: int, device, source: str = "os", generator: torch.Generator | None = None, nonzero: bool = False) -> torch.Tensor:
    """Uniform integers in [0, n) (or [1, n)) as [count][nwords(n)] by rejection sampling on word arrays."""
    bits = n.bit_length()
    nw = (bits + 31) // 32
    n_words = np.frombuffer(n.to_bytes(4 * nw, "little"), dtype="<u4")
    out = np.zeros((count, nw), dtype="<u4")
    filled = 0
    while filled < count:
        need = count - filled
        draw = max(16, int(need * 2.2))
        if source == "os":
            cand = random_bits(bits, (draw,), "cpu", "os").numpy().view(np.uint32)
        else:   # the seeded generator draws on its own device; the rejection step runs on the host
            gdev = generator.device if generator is not None else "cpu"
            cand = random_bits(bits, (draw,), gdev, "torch", generator).cpu().numpy().view(np.uint32)
        lt = np.zeros(draw, dtype=bool)
        eq = np.ones(draw, dtype=bool)
        for k in range(nw - 1, -1, -1):
            lt |= eq & (cand[:, k] < n_words[k])
            eq &= cand[:, k] == n_words[k]
        ok = lt
        if nonzero:
            ok &= cand.any(axis=1)
        good = cand[ok][:need]
        out[filled:filled + len(good)] = good
        filled += len(good)
    return torch.from_numpy(out.view(np.int32)).to(device)


def random_permutations(count: int, k: int, device, source: str = "os", generator: torch.Generator | None = None) -> torch.Tensor:
    """`count` independent uniform permutations of range(k) as int64 [count][k] (argsort of 64-bit random keys)."""
    if source == "os":
        keys = np.frombuffer(os.urandom(8 * count * k), dtype="<u8").reshape(count, k)
        return torch.from_numpy(np.argsort(keys, axis=1, kind="stable").astype(np.int64)).to(device)
    keys = torch.rand((count, k), generator=generator, device=device, dtype=torch.float64)
    return torch.argsort(keys, dim=1)


def random_bits_u64(count: int, device, source: str = "os", generator: torch.Generator | None = None) -> torch.Tensor:
    """One uniform bit per item as int64 [count] (step 4g's delta_A)."""
    if source == "os":
        b = np.frombuffer(os.urandom(count), dtype="u1") & 1
        return torch.from_numpy(b.astype(np.int64)).to(device)
    return torch.randint(0, 2, (count,), generator=generator, device=device, dtype=torch.int64)
