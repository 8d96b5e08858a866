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
            _watch(self, loop, fut)
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
    on a byte transport (wire.py) and JSON scheme documents -- and the ONE-comparison protocol's messages: the scheme pair travels as
    its public document, ciphertexts / lists / tuples of them in wire.pack_session_message's form once `bind_schemes` has named
    the schemes they belong to (Initiator and KeyHolder do that themselves), so that the reference's call shape -- many concurrent
    `perform_secure_comparison` sessions, SC/initiator.py:69-175 -- runs between two processes, each coalescing its own sessions'
    steps into batch launches (coalesce.py)."""

    device_tensors = False

    def __init__(self, sock, alloc=None, timeout_s: float | None = 600.0, max_payload: int = 1 << 32, max_pending: int = 1 << 34,
                 engine=None) -> None:
        """max_payload: the largest frame accepted from the peer (the largest batch message of BASELINE's configurations is 1.1 GB);
        max_pending: the most bytes held for messages nobody has asked for yet.  Both bound what a misbehaving peer can make this
        process allocate: a frame beyond them closes the connection."""
        sock.setblocking(False)
        self.sock, self.timeout_s = sock, timeout_s
        self.max_payload, self.max_pending = int(max_payload), int(max_pending)
        self._pending = 0
        self.engine = engine                 # for the scheme objects made of a received scheme document (None: the default engine)
        self._schemes = None
        self._scheme_docs: dict[str, Any] = {}
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

    def bind_schemes(self, paillier, dgk) -> None:
        """The schemes whose ciphertexts this connection carries for the ONE-comparison protocol (wire.pack_session_message): objects
        that arrive are bound to them.  Initiator and KeyHolder call this themselves once they hold their schemes."""
        self._schemes = (paillier, dgk)

    async def _read_exact(self, loop, view: memoryview) -> None:
        got = 0
        while got < len(view):
            n = await loop.sock_recv_into(self.sock, view[got:])
            if n == 0:
                raise ConnectionError("peer closed the connection")
            got += n

    async def _read_frames(self) -> None:
        """Frames out of a read buffer filled 256 KB at a time (a burst of a thousand sessions is thousands of frames of a few KB:
        one wake-up per chunk, not three per frame); a payload of a megabyte or more -- a batch message -- is received straight into a
        buffer from `alloc`, smaller ones are delivered as `bytes` of their own."""
        loop = asyncio.get_running_loop()
        buf, off = bytearray(), 0
        chunk = memoryview(bytearray(1 << 18))

        async def fill(need: int) -> None:
            nonlocal buf, off
            while len(buf) - off < need:
                if off > (1 << 16) and 2 * off > len(buf):
                    del buf[:off]
                    off = 0
                n = await loop.sock_recv_into(self.sock, chunk)
                if n == 0:
                    raise ConnectionError("peer closed the connection")
                buf += chunk[:n]

        try:
            while True:
                await fill(4)
                (n,) = struct.unpack_from("<I", buf, off)
                if n > 4096:
                    raise ValueError("malformed frame (message id length)")
                await fill(n + 12)
                msg_id = bytes(buf[off + 4:off + 4 + n]).decode()
                (size,) = struct.unpack_from("<Q", buf, off + 4 + n)
                off += n + 12
                if size > self.max_payload:
                    raise ValueError(f"frame {msg_id!r} announces {size} bytes, more than max_payload = {self.max_payload}")
                if msg_id in self._box:
                    raise ValueError(f"second message {msg_id!r} while the first is still unclaimed")
                if msg_id not in self._waiters and self._pending + size > self.max_pending:
                    raise ValueError(f"more than max_pending = {self.max_pending} bytes of unclaimed messages")
                if size < _DIRECT:
                    await fill(size)
                    payload = bytes(buf[off:off + size])
                    off += size
                else:
                    payload = self._alloc(size)
                    view = memoryview(payload).cast("B")
                    have = min(len(buf) - off, size)
                    view[:have] = buf[off:off + have]
                    off += have
                    await self._read_exact(loop, view[have:])
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

    def _encode(self, message: Any):
        """Bytes as they are; the scheme pair of `make_and_send_encryption_schemes` as its public document; ciphertexts, lists and tuples
        of them in wire.py's form for the one-comparison protocol (after what every serializing transport does to them: a ciphertext
        that is not fresh is randomized, with a warning, and the sender's copy stops being fresh)."""
        try:
            return memoryview(message).cast("B")
        except TypeError:
            pass
        from . import wire

        if type(message) is tuple and len(message) == 2 and all(hasattr(m, "public_key") for m in message):
            return _SCHEMES_MAGIC + wire.pack_public_schemes(*message)
        if self._schemes is None:
            raise TypeError(f"StreamCommunicator carries bytes and, once bind_schemes() has named the schemes, ciphertexts; not {type(message).__name__}")
        return wire.pack_session_message(_as_on_wire(message), *self._schemes)

    def _decode(self, payload: Any) -> Any:
        view = memoryview(payload).cast("B")
        head = bytes(view[:4])
        if head == _OBJ_MAGIC:
            if self._schemes is None:
                raise ValueError("a ciphertext message arrived before bind_schemes() named this party's schemes")
            from . import wire

            return wire.unpack_session_message(payload, *self._schemes)
        if head == _SCHEMES_MAGIC:
            doc = bytes(view[4:])
            if self._scheme_docs.get("doc") != doc:          # every session announces the schemes again: one pair of objects per document
                from . import wire

                self._scheme_docs = {"doc": doc, "schemes": wire.unpack_public_schemes(doc, self.engine)}
            return self._scheme_docs["schemes"]
        return payload

    async def send(self, party_id: str, message: Any, msg_id: str) -> None:
        payload = self._encode(message)
        loop = asyncio.get_running_loop()
        if self._send_lock is None:
            self._send_lock = asyncio.Lock()
        ident = msg_id.encode()
        head = struct.pack("<I", len(ident)) + ident + struct.pack("<Q", len(payload))
        async with self._send_lock:                      # frames of concurrent sub-sessions must not interleave
            if len(payload) < (1 << 16):
                await loop.sock_sendall(self.sock, head + bytes(payload))
            else:
                await loop.sock_sendall(self.sock, head)
                await loop.sock_sendall(self.sock, payload)

    async def recv(self, party_id: str, msg_id: str) -> Any:
        if self._pump is None:
            self._pump = asyncio.ensure_future(self._read_frames())
        if msg_id in self._box:
            payload = self._box.pop(msg_id)
            self._pending -= len(memoryview(payload).cast("B"))
            return self._decode(payload)
        if self._closed is not None:
            raise ConnectionError(f"connection closed: {self._closed!r}")
        loop = asyncio.get_running_loop()
        fut = loop.create_future()
        self._waiters[msg_id] = fut
        if self.timeout_s is not None:                   # one watchdog per connection, not a timer per receive
            fut._sc_deadline = loop.time() + self.timeout_s        # type: ignore[attr-defined]
            fut._sc_what = (msg_id, party_id)                      # type: ignore[attr-defined]
            _watch(self, loop, fut)
        try:
            return self._decode(await fut)
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


_DIRECT = 1 << 20            # payloads from this size on are received straight into a buffer from `alloc`
_OBJ_MAGIC, _SCHEMES_MAGIC = b"SCO1", b"SCS1"


def _watch(comm, loop, fut) -> None:
    if comm.__dict__.get("_watch_loop") is not loop:
        comm._watched, comm._watch_loop = [], loop
        loop.call_later(min(1.0, max(0.01, comm.timeout_s)), _check_deadlines, comm, loop)
    comm._watched.append(fut)


def _check_deadlines(comm, loop) -> None:
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
