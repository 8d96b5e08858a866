"""Message transport between the two players.

Any object with awaitable ``send(party_id, message, msg_id)`` and ``recv(party_id, msg_id)`` works (structural typing,
as in the reference's communicator module); the inter-party HTTP transport itself is out of scope here.
``InMemoryCommunicator`` is a minimal single-process transport for demos and tests: both players share one mailbox
and messages are keyed by their ``msg_id`` (the session-numbered step names keep concurrent runs apart).
"""
from __future__ import annotations

import asyncio
import struct
from typing import Any, Protocol, runtime_checkable


@runtime_checkable
class Communicator(Protocol):
    """Structural type of a transport usable by Initiator and KeyHolder."""

    async def send(self, party_id: str, message: Any, msg_id: str) -> None:
        """Deliver `message` to `party_id` under the label `msg_id`."""

    async def recv(self, party_id: str, msg_id: str) -> Any:
        """Wait for the message labelled `msg_id` from `party_id`."""


_WAITERS = "\0waiting receivers"    # reserved mailbox key: msg_id -> future of a receiver that arrived before its message


class InMemoryCommunicator:
    """Shared-mailbox transport for two players living in one process.  Event-driven: a receiver that arrives before its message
    leaves a future in the mailbox and the sender resolves it -- no polling, so a thousand concurrent sessions waiting for
    their peers cost the event loop nothing (round 4 polled with `asyncio.sleep(0)`: every waiting session woke on every turn
    of the loop)."""

    def __init__(self, mailbox: dict[str, Any] | None = None, max_polls: int = 1_000_000, device_tensors: bool = True,
                 timeout_s: float | None = 600.0) -> None:
        self.mailbox: dict[str, Any] = {} if mailbox is None else mailbox
        self.max_polls = max_polls                   # (kept for callers of the polling form; unused)
        self.timeout_s = timeout_s
        # Both endpoints live in one process: a batch message may carry the device arrays themselves (wire.DeviceArrays) instead
        # of bytes.  False makes the batch protocol serialize as it would for a real transport (one pinned host buffer).
        self.device_tensors = device_tensors

    def _watch(self, loop, fut) -> None:
        if self.__dict__.get("_watch_loop") is not loop:
            self._watched, self._watch_loop = [], loop
            loop.call_later(min(1.0, max(0.01, self.timeout_s)), _check_deadlines, self, loop)
        self._watched.append(fut)

    def peer(self) -> "InMemoryCommunicator":
        """A second endpoint on the same mailbox (hand it to the other player)."""
        return InMemoryCommunicator(self.mailbox, self.max_polls, self.device_tensors, self.timeout_s)

    async def send(self, party_id: str, message: Any, msg_id: str) -> None:
        if msg_id in self.mailbox:
            raise RuntimeError(f"message id {msg_id!r} is already pending")
        payload = _as_on_wire(message)
        waiters = self.mailbox.get(_WAITERS)
        fut = waiters.pop(msg_id, None) if waiters else None
        if waiters is not None and not waiters:
            del self.mailbox[_WAITERS]
        if fut is None or fut.done():                # nobody waits yet (or the receiver gave up): the message waits
            self.mailbox[msg_id] = payload
            return
        loop = fut.get_loop()
        if loop is asyncio.get_running_loop():
            fut.set_result(payload)
        else:                                        # the receiver lives on another thread's event loop
            loop.call_soon_threadsafe(lambda: fut.done() or fut.set_result(payload))

    async def recv(self, party_id: str, msg_id: str) -> Any:
        if msg_id in self.mailbox:
            return self.mailbox.pop(msg_id)
        loop = asyncio.get_running_loop()
        fut = loop.create_future()
        waiters = self.mailbox.setdefault(_WAITERS, {})
        if msg_id in waiters:
            raise RuntimeError(f"two receivers wait for message {msg_id!r}")
        waiters[msg_id] = fut
        if self.timeout_s is not None:
            # one watchdog per endpoint and event loop instead of a timer per receive (a thousand sessions wait at once)
            fut._sc_deadline = loop.time() + self.timeout_s        # type: ignore[attr-defined]
            fut._sc_what = (msg_id, party_id)                      # type: ignore[attr-defined]
            self._watch(loop, fut)
        try:
            return await fut
        finally:
            left = self.mailbox.get(_WAITERS)
            if left is not None and left.get(msg_id) is fut:
                del left[msg_id]
                if not left:
                    del self.mailbox[_WAITERS]


class StreamCommunicator:
    """Bytes over one asyncio stream pair (a Unix or TCP socket): the minimal transport between two PROCESSES -- the reference's
    players live in separate processes or hosts (SC/test/integration/test_pool.py:41-73, over tno.mpc.communication's HTTP pools,
    which are out of scope here).  Frames are `id length u32 | id | payload length u64 | payload`; a reader task files arriving
    frames under their message ids, so concurrent sub-sessions (chunked batches) share the connection.  Carries what the batch
    protocol puts on a byte transport (wire.py) and JSON scheme documents; ciphertext OBJECTS of the single-comparison protocol need
    a serializer of their own and are refused."""

    device_tensors = False

    def __init__(self, reader: asyncio.StreamReader, writer: asyncio.StreamWriter, timeout_s: float | None = 600.0) -> None:
        self.reader, self.writer, self.timeout_s = reader, writer, timeout_s
        self._box: dict[str, Any] = {}
        self._waiters: dict[str, asyncio.Future] = {}
        self._pump: asyncio.Task | None = None
        self._closed: BaseException | None = None

    async def _read_frames(self) -> None:
        try:
            while True:
                (n,) = struct.unpack("<I", await self.reader.readexactly(4))
                if n > 4096:
                    raise ValueError("malformed frame (message id length)")
                msg_id = (await self.reader.readexactly(n)).decode()
                (size,) = struct.unpack("<Q", await self.reader.readexactly(8))
                payload = await self.reader.readexactly(size)
                fut = self._waiters.pop(msg_id, None)
                if fut is not None and not fut.done():
                    fut.set_result(payload)
                else:
                    self._box[msg_id] = payload
        except (asyncio.IncompleteReadError, ConnectionError, ValueError) as exc:
            self._closed = exc
            for fut in self._waiters.values():
                if not fut.done():
                    fut.set_exception(ConnectionError(f"connection closed while waiting: {exc!r}"))
            self._waiters.clear()

    async def send(self, party_id: str, message: Any, msg_id: str) -> None:
        try:
            payload = memoryview(message).cast("B")
        except TypeError:
            raise TypeError(f"StreamCommunicator carries bytes (batch messages, scheme documents), not {type(message).__name__}") from None
        ident = msg_id.encode()
        self.writer.write(struct.pack("<I", len(ident)) + ident + struct.pack("<Q", len(payload)))
        self.writer.write(payload)
        await self.writer.drain()

    async def recv(self, party_id: str, msg_id: str) -> Any:
        if self._pump is None:
            self._pump = asyncio.ensure_future(self._read_frames())
        if msg_id in self._box:
            return self._box.pop(msg_id)
        if self._closed is not None:
            raise ConnectionError(f"connection closed: {self._closed!r}")
        fut = asyncio.get_running_loop().create_future()
        self._waiters[msg_id] = fut
        try:
            return await (fut if self.timeout_s is None else asyncio.wait_for(fut, self.timeout_s))
        except asyncio.TimeoutError:
            raise TimeoutError(f"no message {msg_id!r} from {party_id!r}") from None
        finally:
            self._waiters.pop(msg_id, None)

    async def close(self) -> None:
        if self._pump is not None:
            self._pump.cancel()
        self.writer.close()
        try:
            await self.writer.wait_closed()
        except Exception:  # noqa: BLE001 -- the peer may be gone already
            pass


def _check_deadlines(comm: "InMemoryCommunicator", loop) -> None:
    now, alive = loop.time(), []
    for fut in comm._watched:
        if fut.done():
            continue
        if now >= fut._sc_deadline:
            msg_id, party_id = fut._sc_what
            fut.set_exception(TimeoutError(f"no message {msg_id!r} from {party_id!r}"))
        else:
            alive.append(fut)
    comm._watched = alive
    if alive:
        loop.call_later(min(1.0, max(0.01, min(f._sc_deadline for f in alive) - now)), _check_deadlines, comm, loop)
    else:
        comm._watch_loop = None


def _as_on_wire(message: Any) -> Any:
    """Mimic what a serializing transport does to ciphertext objects (the reference's transports randomize a non-fresh
    ciphertext, with a warning, when it is serialized): nested tuples / lists are walked, other payloads pass through."""
    fw = getattr(message, "for_wire", None)
    if fw is not None:
        return fw()
    kind = type(message)
    if kind is list:
        if message and hasattr(type(message[0]), "wire_list"):
            return type(message[0]).wire_list(message)           # a list of ciphertexts of one scheme: one pass, one look-up of the public scheme
        return [_as_on_wire(m) for m in message]
    if kind is tuple:
        return tuple(_as_on_wire(m) for m in message)
    return message
