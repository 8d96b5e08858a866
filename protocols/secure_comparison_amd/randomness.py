"""The random inputs of a batch, drawn on the device.

`source="device"` (default) uses the library's counter-mode CSPRNG -- ChaCha20 blocks keyed per engine with 32 bytes from
the operating system (csrc/sc_rng.h, Engine.rng_*): rejection sampling below N / u, the coins and the Fisher-Yates shuffles
all run on the GPU, so a batch of 65536 comparisons never moves its ~0.3 GB of draws over PCIe (the reference draws each value
with `secrets`, SC/initiator.py:223, :250, :420, :512).  `source="torch"` uses a seeded torch generator on the engine's device
-- NOT cryptographic, for benchmarks and reproducible tests only; every helper honours it, so
boot_pools(source="torch", generator=g) is reproducible."""
from __future__ import annotations

import torch


def _device(engine):
    return engine.device if hasattr(engine, "device") and not isinstance(engine, (str, torch.device)) else torch.device(engine)


def _need_engine(engine, what: str):
    if not hasattr(engine, "rng_bits"):
        raise ValueError(f"{what}: source='device' needs the engine whose generator draws (got {engine!r})")
    return engine


def _torch_words(shape: tuple[int, ...], device, generator) -> torch.Tensor:
    return torch.randint(-(2 ** 31), 2 ** 31, shape, generator=generator, device=device, dtype=torch.int64).to(torch.int32)


def random_bits(bits: int, lead_shape: tuple[int, ...], engine, source: str = "device", generator: torch.Generator | None = None) -> torch.Tensor:
    """Uniform integers below 2^bits as word arrays [*lead_shape][ceil(bits/32)] (int32 view of uint32)."""
    nw = (bits + 31) // 32
    count = 1
    for d in lead_shape:
        count *= d
    if source == "device":
        return _need_engine(engine, "random_bits").rng_bits(bits, count).reshape(tuple(lead_shape) + (nw,))
    if source != "torch":
        raise ValueError(f"unknown randomness source {source!r}")
    w = _torch_words(tuple(lead_shape) + (nw,), generator.device if generator is not None else _device(engine), generator).to(_device(engine))
    top = bits - 32 * (nw - 1)
    if top < 32:
        w[..., -1] &= (1 << top) - 1
    return w


def _below_mask(cand: torch.Tensor, n: int, nonzero: bool) -> torch.Tensor:
    """bool [rows]: candidate (words, least significant first) < n (and != 0): the most significant differing word decides."""
    nw = cand.shape[-1]
    u = cand.to(torch.int64) & 0xFFFFFFFF
    lt = torch.zeros(cand.shape[0], dtype=torch.bool, device=cand.device)
    eq = torch.ones_like(lt)
    for k in range(nw - 1, -1, -1):
        nk = (n >> (32 * k)) & 0xFFFFFFFF
        lt |= eq & (u[:, k] < nk)
        eq &= u[:, k] == nk
    return lt & (u != 0).any(dim=1) if nonzero else lt


def uniform_below(n: int, count: int, engine, source: str = "device", generator: torch.Generator | None = None, nonzero: bool = False) -> torch.Tensor:
    """Uniform integers in [0, n) (or [1, n)) as [count][nwords(n)], by rejection sampling on the device."""
    if source == "device":
        return _need_engine(engine, "uniform_below").rng_below(n, count, nonzero)
    bits = n.bit_length()
    nw = (bits + 31) // 32
    dev = _device(engine)
    out = torch.empty((count, nw), dtype=torch.int32, device=dev)
    filled = 0
    while filled < count:          # every candidate is accepted with probability > 1/2: two or three rounds
        need = count - filled
        cand = random_bits(bits, (max(16, int(need * 2.2)),), engine, "torch", generator)
        good = cand[_below_mask(cand, n, nonzero)][:need]
        out[filled:filled + good.shape[0]] = good
        filled += good.shape[0]
    return out


def random_permutations(count: int, k: int, engine, source: str = "device", generator: torch.Generator | None = None) -> torch.Tensor:
    """`count` independent uniform permutations of range(k) as int64 [count][k]."""
    if source == "device":
        return _need_engine(engine, "random_permutations").rng_permutations(k, count)
    if source != "torch":
        raise ValueError(f"unknown randomness source {source!r}")
    keys = torch.rand((count, k), generator=generator, device=generator.device if generator is not None else _device(engine), dtype=torch.float64)
    return torch.argsort(keys, dim=1).to(_device(engine))


def random_coins(count: int, engine, source: str = "device", generator: torch.Generator | None = None) -> torch.Tensor:
    """One uniform bit per item as int64 [count] (step 4g's delta_A, SC/initiator.py:420)."""
    if source == "device":
        return _need_engine(engine, "random_coins").rng_coins(count)
    if source != "torch":
        raise ValueError(f"unknown randomness source {source!r}")
    dev = _device(engine)
    return torch.randint(0, 2, (count,), generator=generator, device=generator.device if generator is not None else dev, dtype=torch.int64).to(dev)
