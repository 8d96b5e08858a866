"""Message transport between the two players.

Any object with awaitable ``send(party_id, message, msg_id)`` and ``recv(party_id, msg_id)`` works (structural typing,
as in the reference's communicator module); the inter-party HTTP transport itself is out of scope here.
``InMemoryCommunicator`` is a minimal single-process transport for demos and tests: both players share one mailbox
and messages are keyed by their ``msg_id`` (the session-numbered step names keep concurrent runs apart).
"""
from __future__ import annotations

import asyncio
from typing import Any, Protocol, runtime_checkable


@runtime_checkable
class Communicator(Protocol):
    """Structural type of a transport usable by Initiator and KeyHolder."""

    async def send(self, party_id: str, message: Any, msg_id: str) -> None:
        """Deliver `message` to `party_id` under the label `msg_id`."""

    async def recv(self, party_id: str, msg_id: str) -> Any:
        """Wait for the message labelled `msg_id` from `party_id`."""


class InMemoryCommunicator:
    """Shared-mailbox transport for two players living in one event loop."""

    def __init__(self, mailbox: dict[str, Any] | None = None, max_polls: int = 1_000_000, device_tensors: bool = True) -> None:
        self.mailbox: dict[str, Any] = {} if mailbox is None else mailbox
        self.max_polls = max_polls
        # Both endpoints live in one process: a batch message may carry the device arrays themselves (wire.DeviceArrays) instead
        # of bytes.  False makes the batch protocol serialize as it would for a real transport (one pinned host buffer).
        self.device_tensors = device_tensors

    def peer(self) -> "InMemoryCommunicator":
        """A second endpoint on the same mailbox (hand it to the other player)."""
        return InMemoryCommunicator(self.mailbox, self.max_polls, self.device_tensors)

    async def send(self, party_id: str, message: Any, msg_id: str) -> None:
        if msg_id in self.mailbox:
            raise RuntimeError(f"message id {msg_id!r} is already pending")
        self.mailbox[msg_id] = _as_on_wire(message)

    async def recv(self, party_id: str, msg_id: str) -> Any:
        for i in range(self.max_polls):
            if msg_id in self.mailbox:
                return self.mailbox.pop(msg_id)
            # a peer that is in the middle of a long GPU step (or whose message is still draining over PCIe) is not polled in a
            # tight loop for its whole duration
            await asyncio.sleep(0 if i < 2000 else 0.0002)
        raise TimeoutError(f"no message {msg_id!r} from {party_id!r}")


def _as_on_wire(message: Any) -> Any:
    """Mimic what a serializing transport does to ciphertext objects (the reference's transports randomize a non-fresh
    ciphertext, with a warning, when it is serialized): nested tuples / lists are walked, other payloads pass through."""
    if hasattr(message, "for_wire"):
        return message.for_wire()
    if isinstance(message, tuple):
        return tuple(_as_on_wire(m) for m in message)
    if isinstance(message, list):
        return [_as_on_wire(m) for m in message]
    return message
