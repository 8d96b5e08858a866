"""Python int <-> little-endian uint32 word arrays (the C ABI's residue format)."""
from __future__ import annotations

from typing import Iterable, Sequence

import numpy as np


def nwords_for_bits(bits: int) -> int:
    return (bits + 31) // 32


def int_to_words(x: int, nwords: int) -> np.ndarray:
    if x < 0 or x >> (32 * nwords):
        raise ValueError("integer does not fit the word array")
    return np.frombuffer(x.to_bytes(4 * nwords, "little"), dtype="<u4").copy()


def ints_to_words(xs: Iterable[int], nwords: int) -> np.ndarray:
    xs = list(xs)
    buf = b"".join(int(x).to_bytes(4 * nwords, "little") for x in xs)
    return np.frombuffer(buf, dtype="<u4").reshape(len(xs), nwords).copy()


def words_to_int(words: Sequence[int] | np.ndarray) -> int:
    return int.from_bytes(np.ascontiguousarray(words, dtype="<u4").tobytes(), "little")


def words_to_ints(arr: np.ndarray) -> list[int]:
    arr = np.ascontiguousarray(arr, dtype="<u4")
    n = arr.shape[-1]
    flat = arr.reshape(-1, n)
    raw = flat.tobytes()
    step = 4 * n
    return [int.from_bytes(raw[i * step:(i + 1) * step], "little") for i in range(flat.shape[0])]


class RowBlock:
    """The rows that belong to ONE session inside a batch array: row j = base[j, b] of a bit-major host array [rows][K][words] (what a
    batched step of K coalesced sessions downloads).  Ciphertext objects of a message refer to such a block and a row number instead of
    holding Python integers; the next batched call recognises the whole batch by its blocks (coalesce.stack_blocks) and takes the
    array back as it is."""

    __slots__ = ("base", "b")

    def __init__(self, base: np.ndarray, b: int) -> None:
        self.base, self.b = base, b

    def __len__(self) -> int:
        return self.base.shape[0]

    @property
    def words(self) -> int:
        return self.base.shape[-1]

    def row(self, j: int) -> np.ndarray:
        return self.base[j, self.b]

    def array(self) -> np.ndarray:
        """[rows][words] (a strided view)."""
        return self.base[:, self.b]
