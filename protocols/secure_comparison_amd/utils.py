"""Bit (de)composition helpers shared by both players (same contract as the reference's utils module,
/root/reference/src/tno/mpc/protocols/secure_comparison/utils.py:6-38)."""
from __future__ import annotations

from typing import Sequence


def to_bits(integer: int, bit_length: int) -> list[int]:
    """Non-negative integer -> `bit_length` bits, least significant first (utils.py:6-21)."""
    assert integer < (1 << bit_length)
    return [(integer >> position) & 1 for position in range(bit_length)]


def from_bits(bits: Sequence[int]) -> int:
    """Bits (least significant first) -> integer (utils.py:24-38); only entries equal to 1 count."""
    return sum(1 << position for position, bit in enumerate(bits) if bit == 1)
