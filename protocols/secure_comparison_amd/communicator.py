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
    """Bytes over one connected socket (Unix or TCP): the minimal transport between two PROCESSES -- the reference's players live in
    separate processes or hosts (SC/test/integration/test_pool.py:41-73, over tno.mpc.communication's HTTP pools, which are out of
    scope here).  Frames are `id length u32 | id | payload length u64 | payload`; a reader task files arriving frames under their
    message ids, so concurrent sub-sessions (chunked batches) share the connection.  Payloads go out with `loop.sock_sendall`
    straight from the sender's buffer (the pinned message the kernels wrote, wire.reserve) and are received with
    `loop.sock_recv_into` into a buffer from `alloc(nbytes)` -- hand in wire.pinned_buffer and a batch message lands in pinned host
    memory, from where the host-to-device copies run on an SDMA engine without a staging copy.  Carries what the batch protocol puts
    on a byte transport (wire.py) and JSON scheme documents; ciphertext OBJECTS of the single-comparison protocol need a serializer
    of their own and are refused."""

    device_tensors = False

    def __init__(self, sock, alloc=None, timeout_s: float | None = 600.0, max_payload: int = 1 << 32, max_pending: int = 1 << 34) -> None:
        """max_payload: the largest frame accepted from the peer (the largest batch message of BASELINE's configurations is 1.1 GB);
        max_pending: the most bytes held for messages nobody has asked for yet.  Both bound what a misbehaving peer can make this
        process allocate: a frame beyond them closes the connection."""
        sock.setblocking(False)
        self.sock, self.timeout_s = sock, timeout_s
        self.max_payload, self.max_pending = int(max_payload), int(max_pending)
        self._pending = 0
        self._alloc = alloc if alloc is not None else bytearray
        self._box: dict[str, Any] = {}
        self._waiters: dict[str, asyncio.Future] = {}
        self._pump: asyncio.Task | None = None
        self._closed: BaseException | None = None
        self._send_lock: asyncio.Lock | None = None

    @classmethod
    async def open_unix(cls, path: str, alloc=None, wait_s: float = 60.0, **options) -> "StreamCommunicator":
        """Connect to a listening Unix socket (waiting for the peer to start listening)."""
        import os
        import socket

        loop, t_end = asyncio.get_running_loop(), asyncio.get_running_loop().time() + wait_s
        while True:
            sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            sock.setblocking(False)
            try:
                if not os.path.exists(path):
                    raise FileNotFoundError(path)
                await loop.sock_connect(sock, path)
                return cls(sock, alloc, **options)
            except (FileNotFoundError, ConnectionRefusedError):
                sock.close()
                if loop.time() > t_end:
                    raise
                await asyncio.sleep(0.05)

    @classmethod
    async def accept_unix(cls, path: str, alloc=None, **options) -> "StreamCommunicator":
        """Listen on a Unix socket and return the communicator of the first connection."""
        import socket

        srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        srv.bind(path)
        srv.listen(1)
        srv.setblocking(False)
        try:
            conn, _ = await asyncio.get_running_loop().sock_accept(srv)
        finally:
            srv.close()
        return cls(conn, alloc, **options)

    async def _read_exact(self, loop, view: memoryview) -> None:
        got = 0
        while got < len(view):
            n = await loop.sock_recv_into(self.sock, view[got:])
            if n == 0:
                raise ConnectionError("peer closed the connection")
            got += n

    async def _read_frames(self) -> None:
        loop = asyncio.get_running_loop()
        head = bytearray(4)
        try:
            while True:
                await self._read_exact(loop, memoryview(head))
                (n,) = struct.unpack("<I", head)
                if n > 4096:
                    raise ValueError("malformed frame (message id length)")
                rest = bytearray(n + 8)
                await self._read_exact(loop, memoryview(rest))
                msg_id = bytes(rest[:n]).decode()
                (size,) = struct.unpack("<Q", rest[n:])
                if size > self.max_payload:
                    raise ValueError(f"frame {msg_id!r} announces {size} bytes, more than max_payload = {self.max_payload}")
                if msg_id in self._box:
                    raise ValueError(f"second message {msg_id!r} while the first is still unclaimed")
                if msg_id not in self._waiters and self._pending + size > self.max_pending:
                    raise ValueError(f"more than max_pending = {self.max_pending} bytes of unclaimed messages")
                payload = self._alloc(size)
                if size:
                    await self._read_exact(loop, memoryview(payload).cast("B"))
                fut = self._waiters.pop(msg_id, None)
                if fut is not None and not fut.done():
                    fut.set_result(payload)
                else:
                    self._box[msg_id] = payload
                    self._pending += size
        except Exception as exc:  # noqa: BLE001 -- whatever ends the reader (the peer, a malformed frame, a failed allocation) ends every wait
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
        loop = asyncio.get_running_loop()
        if self._send_lock is None:
            self._send_lock = asyncio.Lock()
        ident = msg_id.encode()
        async with self._send_lock:                      # frames of concurrent sub-sessions must not interleave
            await loop.sock_sendall(self.sock, struct.pack("<I", len(ident)) + ident + struct.pack("<Q", len(payload)))
            if len(payload):
                await loop.sock_sendall(self.sock, payload)

    async def recv(self, party_id: str, msg_id: str) -> Any:
        if self._pump is None:
            self._pump = asyncio.ensure_future(self._read_frames())
        if msg_id in self._box:
            payload = self._box.pop(msg_id)
            self._pending -= len(memoryview(payload).cast("B"))
            return payload
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
            try:
                await self._pump
            except (asyncio.CancelledError, Exception):  # noqa: BLE001
                pass
        self.sock.close()


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
