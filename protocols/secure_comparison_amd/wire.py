"""Wire format for ciphertext batches.

The reference sends Python scheme / ciphertext objects through its transport's serializer, one object per
ciphertext (SC/test/conftest.py:182,198).  A batch of B comparisons moves ~19 KB per comparison (0.55 GB for each per-bit
vector at B = 65536), so a batch travels in one of two forms, chosen by what the transport can carry:

* **device hand-over** -- a transport whose two endpoints live in one process on one GPU (`communicator.device_tensors` is
  true: InMemoryCommunicator) gets a `DeviceArrays` message: the device tensors themselves plus an event recorded on the
  sender's stream, which the receiver's stream waits for.  No byte ever leaves HBM.
* **bytes** -- any other transport gets ONE buffer per message: header(s) and payload(s) are written into a single pinned host
  buffer by one device-to-host copy per array (no intermediate numpy copies, no concatenation), and handed over as a
  `memoryview`:

      message  = count u32 | { length u64 | array }*            (pack_many)
      array    = magic "SCB1" | dtype code u8 | ndim u8 | pad u16 | dims (ndim x u64) | pad bytes | payload (little-endian words)

  (`pad` is 0 in packed messages; a message whose payloads the producing KERNELS write -- `reserve`, below -- pads so that every
  payload starts on a 256-byte boundary of the pinned buffer.)

  The receiver wraps the payload in place (`torch.frombuffer`) and issues one host-to-device copy per array.

**No device-to-host copy at all (round 4).**  On this runtime every device-to-host `hipMemcpyAsync` is executed by a blit kernel
(`__amd_rocclr_copyBuffer`, 4096 waves for a 0.3 GB message; host-to-device copies use an SDMA engine -- `tools/gpu_copy_probe.py`):
beside another session's chip-filling launches it waits for wave slots, and once running it holds them for the ~5 ms the bytes
take to cross PCIe.  So the sender does not copy: `reserve` lays the message out in a pooled pinned buffer up front and hands back
its payload regions as arrays; the step's LAST launch stores its results straight into them (pinned host memory is mapped into
the device's address space; the stores are posted PCIe writes spread over the launch), and `Reserved.finish` only waits for the
compute stream's event.

**Pipelining (round 4).**  A byte message of a batch is ~0.5 GB; copying it out of HBM takes ~10 ms during which a single
session's GPU would idle.  `outgoing_async` therefore issues the device-to-host copies on a COPY STREAM of its own (after an event
on the compute stream) and returns an awaitable: while one chunk's message drains over PCIe the event loop runs the next
chunk's step (`Initiator.perform_secure_comparison_batch(chunks=k)` cuts a batch into k sub-sessions that share the connection;
the key holder learns the plan from the first message, `pack_plan`).  `incoming` does the mirror image: host-to-device copies on
the copy stream, the compute stream waits for their event.

Either way nothing in a message is trusted: shapes and dtypes are checked against the bytes that arrived and then against
what this party's own parameters (l, B, key sizes) dictate (`expect_array`).  Public keys travel as a small JSON document
(integers as hex).  `STATS` accumulates the seconds and bytes spent packing / unpacking (bench.py reports the split)."""
from __future__ import annotations

import json
import struct
import threading
import time
from dataclasses import dataclass
from typing import Any, Sequence

import numpy as np
import torch

MAGIC = b"SCB1"
_DTYPES = {0: (torch.int32, np.dtype("<i4")), 1: (torch.int64, np.dtype("<i8")), 2: (torch.uint8, np.dtype("u1"))}
_CODES = {t: c for c, (t, _) in _DTYPES.items()}
MAX_NDIM = 4
MAX_SCHEME_DOCUMENT = 1 << 16
MAX_WINDOW_FROM_PEER = 16
STATS = {"pack_s": 0.0, "unpack_s": 0.0, "bytes": 0, "device_arrays": 0}


def reset_stats() -> None:
    STATS.update(pack_s=0.0, unpack_s=0.0, bytes=0, device_arrays=0)


@dataclass
class DeviceArrays:
    """A batch message that never leaves the GPU: the arrays themselves, to be treated as read-only by the receiver, and the
    event after which they are complete."""

    arrays: tuple[torch.Tensor, ...]
    ready: Any = None          # torch.cuda.Event recorded on the sender's stream (None for CPU tensors)


def carries_device_arrays(communicator: Any) -> bool:
    return bool(getattr(communicator, "device_tensors", False))


def _header(t: torch.Tensor) -> bytes:
    if t.dtype not in _CODES:
        raise ValueError(f"unsupported dtype {t.dtype}")
    if t.dim() > MAX_NDIM:
        raise ValueError(f"at most {MAX_NDIM} dimensions")
    return MAGIC + struct.pack("<BBH", _CODES[t.dtype], t.dim(), 0) + struct.pack(f"<{t.dim()}Q", *t.shape)


class _PinnedPool:
    """Pinned staging buffers for outgoing messages, reused from batch to batch.  Allocating pinned memory is slow (tens of
    milliseconds for a 0.3 GB message, measured with tools/gpu_wire_probe.py, and it stalls the other sessions' launches too), and
    torch's caching host allocator does not hand a block back while any stream event on it is pending; a message of the same size is
    packed once per batch, so a handful of buffers serve a long-running session.  A buffer is free again when nothing refers to the
    view that was handed out for its previous message (the transport, the receiver's in-place wrapped arrays and its pending
    host-to-device copies all hold on to that view)."""

    def __init__(self) -> None:
        self._lock = threading.Lock()
        self._bufs: list = []          # [pinned tensor, weakref to the last view handed out or None]

    def take(self, nbytes: int) -> tuple[torch.Tensor, np.ndarray]:
        import weakref

        with self._lock:
            best = None
            for entry in self._bufs:
                if entry[0].numel() >= nbytes and (entry[1] is None or entry[1]() is None):
                    if best is None or entry[0].numel() < best[0].numel():
                        best = entry
            if best is None or best[0].numel() > 2 * nbytes + (1 << 20):
                cap = (nbytes + (1 << 20) - 1) >> 20 << 20          # whole MiB: chunks of nearly equal sizes share buffers
                best = [torch.empty(cap, dtype=torch.uint8, pin_memory=True), None]
                self._bufs.append(best)
                if len(self._bufs) > 64:                             # never an unbounded cache: drop the free ones
                    self._bufs = [e for e in self._bufs if e is best or (e[1] is not None and e[1]() is not None)]
            view = best[0][:nbytes].numpy()
            best[1] = weakref.ref(view)
            return best[0][:nbytes], view


_pinned_pool = _PinnedPool()


def pinned_buffer(nbytes: int):
    """A writable buffer of `nbytes` bytes of PINNED host memory from the message pool (returned to the pool when nothing refers to
    it any more): what a receiving transport reads a batch message into, so that the host-to-device copies of `incoming` need no
    staging copy (communicator.StreamCommunicator(alloc=wire.pinned_buffer))."""
    if nbytes < (1 << 16):
        return bytearray(nbytes)                      # scheme documents, plans: not worth a pinned block
    return _pinned_pool.take(nbytes)[1]
_copy_streams: dict = {}


def copy_stream(device: torch.device) -> "torch.cuda.Stream":
    """The stream this thread's message copies run on (one per thread and device; never the compute stream)."""
    key = (threading.get_ident(), torch.device(device).index)
    if key not in _copy_streams:
        _copy_streams[key] = torch.cuda.Stream(device=device)
    return _copy_streams[key]


class PendingMessage:
    """A byte message whose payload copies are still in flight on the copy stream.  `await msg.wait()` yields to the event loop
    until they are done and returns the buffer; the source arrays are kept alive until then."""

    def __init__(self, raw: np.ndarray, event, keep: tuple, started: float, begin=None) -> None:
        self._raw, self._event, self._keep, self._t0, self._begin = raw, event, keep, started, begin

    def done(self) -> bool:
        return self._event is None or self._event.query()

    def result(self) -> memoryview:
        """Blocking form."""
        if self._event is not None:
            self._event.synchronize()
        return self._finish()

    async def wait(self) -> memoryview:
        import asyncio

        spins = 0
        while not self.done():
            spins += 1
            await asyncio.sleep(0 if spins < 200 else 0.0001)
        return self._finish()

    def _finish(self) -> memoryview:
        if self._keep is not None:
            # the copies' own duration on the copy stream (they overlap with whatever the compute stream ran meanwhile)
            STATS["pack_s"] += (self._begin.elapsed_time(self._event) * 1e-3) if self._begin is not None else (time.perf_counter() - self._t0)
            self._keep, self._event, self._begin = None, None, None
        return memoryview(self._raw)


def _pack(tensors: Sequence[torch.Tensor], framed: bool, asynchronous: bool = False):
    """One host buffer for the whole message; every payload lands in it by a single copy from wherever the array lives.
    asynchronous: device payloads are copied on the copy stream and a PendingMessage is returned instead of the buffer."""
    _reap()
    on_gpu = [t for t in tensors if t.is_cuda]
    if on_gpu and not asynchronous:  # the copies below wait for the producing kernels anyway; waiting here keeps STATS about the wire only
        torch.cuda.current_stream(on_gpu[0].device).synchronize()
    t0 = time.perf_counter()
    heads = [_header(t) for t in tensors]
    sizes = [len(h) + t.numel() * t.element_size() for h, t in zip(heads, tensors)]
    total = (4 + sum(8 + s for s in sizes)) if framed else sizes[0]
    if on_gpu:
        buf, raw = _pinned_pool.take(total)
    else:
        buf = torch.empty(total, dtype=torch.uint8)
        raw = buf.numpy()
    off = 0
    if framed:
        raw[0:4] = np.frombuffer(struct.pack("<I", len(tensors)), dtype=np.uint8)
        off = 4
    side, keep, begin = None, [], None
    if on_gpu and asynchronous:
        compute = torch.cuda.current_stream(on_gpu[0].device)
        side = copy_stream(on_gpu[0].device)
        side.wait_stream(compute)                     # the arrays are complete once everything queued so far has run
        begin = torch.cuda.Event(enable_timing=True)
        begin.record(side)
    for h, t, s in zip(heads, tensors, sizes):
        if framed:
            raw[off:off + 8] = np.frombuffer(struct.pack("<Q", s), dtype=np.uint8)
            off += 8
        raw[off:off + len(h)] = np.frombuffer(h, dtype=np.uint8)
        nbytes = s - len(h)
        if nbytes:
            # the payload region viewed as bytes of the array's own dtype: one (device-to-)host copy, straight into place
            src = t.detach().contiguous().reshape(-1).view(torch.uint8)
            if side is not None and t.is_cuda:
                with torch.cuda.stream(side):
                    buf[off + len(h):off + s].copy_(src, non_blocking=True)
                src.record_stream(side)
                keep.append(src)
            else:
                buf[off + len(h):off + s].copy_(src, non_blocking=False)
        off += s
    STATS["bytes"] += total
    if asynchronous:
        ev = None
        if side is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(side)
        return PendingMessage(raw, ev, (buf, tuple(keep)), t0, begin)
    STATS["pack_s"] += time.perf_counter() - t0
    return memoryview(raw)


class Reserved:
    """A byte message under construction whose payloads are written by the producing kernels themselves (see the module text):
    `arrays[i]` is the i-th payload as a pinned host array of the requested shape -- pass it as the step's output array --, and
    `await finish()` yields the finished message once everything queued on the current stream has run."""

    ALIGN = 256

    def __init__(self, device, shapes: Sequence[tuple[int, ...]], dtype: torch.dtype = torch.int32) -> None:
        item = torch.empty((), dtype=dtype).element_size()
        code = _CODES[dtype]
        off, plan = 4, []
        for shape in shapes:
            if len(shape) > MAX_NDIM:
                raise ValueError(f"at most {MAX_NDIM} dimensions")
            n = 1
            for d in shape:
                n *= int(d)
            head = 8 + 8 * len(shape)
            start = off + 8 + head                               # where the payload would begin without padding
            pad = (-start) % self.ALIGN
            plan.append((off, head, pad, n * item, tuple(int(d) for d in shape)))
            off = start + pad + n * item
        self._device = torch.device(device)
        self._buf, self._raw = _pinned_pool.take(off)
        raw = self._raw
        raw[0:4] = np.frombuffer(struct.pack("<I", len(plan)), dtype=np.uint8)
        self.arrays = []
        for o, head, pad, nbytes, shape in plan:
            raw[o:o + 8] = np.frombuffer(struct.pack("<Q", head + pad + nbytes), dtype=np.uint8)
            raw[o + 8:o + 8 + head] = np.frombuffer(MAGIC + struct.pack("<BBH", code, len(shape), pad) + struct.pack(f"<{len(shape)}Q", *shape), dtype=np.uint8)
            raw[o + 8 + head:o + 8 + head + pad] = 0
            p0 = o + 8 + head + pad
            self.arrays.append(self._buf[p0:p0 + nbytes].view(dtype).reshape(shape))
        STATS["bytes"] += off

    async def finish(self) -> memoryview:
        import asyncio

        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self._device))
        spins = 0
        while not ev.query():
            spins += 1
            await asyncio.sleep(0 if spins < 200 else 0.0001)
        self.arrays = None
        return memoryview(self._raw)


def reserve(communicator: Any, device, *shapes: tuple[int, ...]) -> "Reserved | None":
    """A message of int32 arrays of these shapes whose payloads the step's kernels will write in place -- or None when the transport
    carries device arrays (then the step allocates device outputs and `outgoing` hands them over) or the arrays live on the CPU."""
    if carries_device_arrays(communicator) or torch.device(device).type != "cuda":
        return None
    return Reserved(device, shapes)


def pack_tensor(t: torch.Tensor) -> memoryview:
    return _pack([t], framed=False)


def pack_many(*tensors: torch.Tensor) -> memoryview:
    return _pack(tensors, framed=True)


def outgoing(communicator: Any, *tensors: torch.Tensor):
    """The message for `tensors` in the form `communicator` carries: the arrays themselves, or one byte buffer."""
    if carries_device_arrays(communicator):
        ev = None
        if tensors and tensors[0].is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(tensors[0].device))
        STATS["device_arrays"] += len(tensors)
        return DeviceArrays(tuple(t.detach() for t in tensors), ev)
    return pack_many(*tensors)


async def outgoing_async(communicator: Any, *tensors: torch.Tensor):
    """`outgoing` for a coroutine: a byte message's device-to-host copies run on the copy stream while the event loop goes on
    with other sub-sessions; resolves to the finished message."""
    if carries_device_arrays(communicator):
        return outgoing(communicator, *tensors)
    return await _pack(tensors, framed=True, asynchronous=True).wait()


PLAN_MAGIC = b"SCPLAN1"


def pack_plan(sizes: Sequence[int]) -> bytes:
    """First message of a chunked batch: how many sub-sessions follow and how many comparisons each carries."""
    return PLAN_MAGIC + json.dumps({"chunks": [int(n) for n in sizes]}).encode()


def plan_of(message: Any) -> list[int] | None:
    """The chunk sizes if `message` is a plan (pack_plan), else None -- then it is the step-1 message of an unchunked batch."""
    if isinstance(message, DeviceArrays):
        return None
    try:
        mv = memoryview(message).cast("B")
    except TypeError:
        return None
    if len(mv) < len(PLAN_MAGIC) or bytes(mv[:len(PLAN_MAGIC)]) != PLAN_MAGIC:
        return None
    try:
        sizes = json.loads(bytes(mv[len(PLAN_MAGIC):]).decode())["chunks"]
    except (ValueError, KeyError, TypeError) as exc:
        raise ValueError("malformed batch plan") from exc
    if not isinstance(sizes, list) or not 1 <= len(sizes) <= 64 or not all(isinstance(n, int) and 0 < n <= 1 << 32 for n in sizes):
        raise ValueError("malformed batch plan")
    return sizes


def incoming(message: Any, device: torch.device | str, expect: int | None, planes_of_one: bool = False) -> list[torch.Tensor]:
    """The arrays of a received message (either form), on `device`; exactly `expect` of them (None: the caller checks the count).
    planes_of_one: the arrays are [B][w] followed by [k][B][w] of the same dtype -- [d] and the planes [beta_i] -- and the caller
    wants them as the consecutive planes of ONE device array (what the initiator's inversion pass takes without a joining copy:
    a 0.3 GB device-to-device copy per batch otherwise, done by a runtime blit kernel that queues behind other sessions' launches)."""
    if isinstance(message, DeviceArrays):
        if expect is not None and len(message.arrays) != expect:
            raise ValueError(f"batch message carries {len(message.arrays)} arrays, expected {expect}")
        dev = torch.device(device)
        for t in message.arrays:
            if not isinstance(t, torch.Tensor) or t.device != dev:
                raise ValueError("device hand-over between different devices: use a byte transport")
        if message.ready is not None:
            torch.cuda.current_stream(dev).wait_event(message.ready)
        return list(message.arrays)
    return unpack_many(message, device, expect, planes_of_one)


def _as_view(buf: Any) -> memoryview:
    try:
        return memoryview(buf).cast("B")
    except TypeError as exc:
        raise ValueError("not a secure-comparison batch message") from exc


_tls = threading.local()   # .inflight: (event, host view, message) of this thread's host-to-device copies that may still read a message's bytes


def _reap() -> None:
    """Let go of the messages whose host-to-device copies have completed (their staging buffers return to the sender's pool)."""
    inflight = _tls.__dict__.get("inflight")
    while inflight and inflight[0][0].query():
        inflight.pop(0)


def unpack_tensor(buf: Any, device: torch.device | str = "cpu", into: torch.Tensor | None = None) -> torch.Tensor:
    """Parse one array message.  The header comes from the peer: everything in it is checked against the bytes that
    actually arrived before an array of that shape is built (ValueError otherwise).  into: a device array of exactly the
    announced dtype and number of elements that receives the payload instead of a new one (returned reshaped as announced)."""
    t0 = time.perf_counter()
    mv = _as_view(buf)
    if len(mv) < 8 or bytes(mv[:4]) != MAGIC:
        raise ValueError("not a secure-comparison batch message")
    code, ndim, pad = struct.unpack_from("<BBH", mv, 4)
    if code not in _DTYPES:
        raise ValueError(f"unknown dtype code {code} in batch message")
    if ndim > MAX_NDIM or pad > 4096 or len(mv) < 8 + 8 * ndim + pad:
        raise ValueError("malformed batch message header")
    shape = struct.unpack_from(f"<{ndim}Q", mv, 8)
    tdt, ndt = _DTYPES[code]
    count = 1
    for d in shape:
        count *= d
    if count * ndt.itemsize != len(mv) - 8 - 8 * ndim - pad:
        raise ValueError(f"batch message announces shape {tuple(shape)} but carries {len(mv) - 8 - 8 * ndim - pad} payload bytes")
    if into is not None and (into.dtype != tdt or into.numel() != count or not into.is_contiguous()):
        raise ValueError(f"batch message announces {count} elements of {tdt}, the receiving array holds {into.numel()} of {into.dtype}")
    if count == 0:
        out = torch.empty(tuple(shape), dtype=tdt, device=device) if into is None else into.reshape(tuple(shape))
    else:
        payload = mv[8 + 8 * ndim + pad:]
        import warnings

        with warnings.catch_warnings():      # immutable `bytes` from a socket: wrapped all the same, and only ever read
            warnings.filterwarnings("ignore", message=".*not writable.*")
            host = torch.frombuffer(payload, dtype=torch.uint8).view(tdt).reshape(tuple(shape))   # wraps the received bytes in place
        dev = torch.device(device)
        if dev.type == "cpu":
            out = host.clone() if into is None else into.reshape(tuple(shape)).copy_(host)
        else:
            # on the copy stream, so that the transfer runs beside the kernels already queued on the compute stream (which then
            # waits for it); from pinned memory the host does not wait either -- the message is kept alive by the array's copy
            compute, side = torch.cuda.current_stream(dev), copy_stream(dev)
            if into is None:
                with torch.cuda.stream(side):
                    out = host.to(dev, non_blocking=True)
                out.record_stream(compute)
            else:
                out = into.reshape(tuple(shape))
                side.wait_stream(compute)                       # `into` was allocated on the compute stream
                with torch.cuda.stream(side):
                    out.copy_(host, non_blocking=True)
                out.record_stream(side)
            ev = torch.cuda.Event()
            ev.record(side)
            compute.wait_event(ev)
            _tls.__dict__.setdefault("inflight", []).append((ev, host, buf))
            _reap()
    STATS["unpack_s"] += time.perf_counter() - t0
    return out


def expect_array(t: torch.Tensor, shape: tuple[int, ...], name: str, dtype: torch.dtype = torch.int32) -> torch.Tensor:
    """A received array must have exactly the dtype and shape this party's own parameters (l, B, key sizes) dictate: the
    kernels take sizes from the local side, never from the message."""
    if not isinstance(t, torch.Tensor):
        raise ValueError(f"{name}: received {type(t).__name__}, expected an array")
    if t.dtype != dtype:
        raise ValueError(f"{name}: received dtype {t.dtype}, expected {dtype}")
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: received shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t.contiguous()


def _peek_shape(mv: memoryview) -> tuple[int, tuple[int, ...]]:
    """(dtype code, shape) announced by an array message, after the same header checks unpack_tensor makes."""
    if len(mv) < 8 or bytes(mv[:4]) != MAGIC:
        raise ValueError("not a secure-comparison batch message")
    code, ndim, pad = struct.unpack_from("<BBH", mv, 4)
    if code not in _DTYPES or ndim > MAX_NDIM or pad > 4096 or len(mv) < 8 + 8 * ndim + pad:
        raise ValueError("malformed batch message header")
    shape = tuple(struct.unpack_from(f"<{ndim}Q", mv, 8))
    count = 1
    for d in shape:
        count *= d
    # the announced shape against the bytes that arrived, BEFORE anyone sizes an array by it (a forged header announcing
    # [255][2^20][2^10] in a 76-byte message would otherwise make the receiver try a terabyte allocation)
    if count * _DTYPES[code][1].itemsize != len(mv) - 8 - 8 * ndim - pad:
        raise ValueError(f"batch message announces shape {shape} but carries {len(mv) - 8 - 8 * ndim - pad} payload bytes")
    return code, shape


def unpack_many(buf: Any, device: torch.device | str = "cpu", expect: int | None = None, planes_of_one: bool = False) -> list[torch.Tensor]:
    mv = _as_view(buf)
    if len(mv) < 4:
        raise ValueError("malformed batch message")
    (n,) = struct.unpack_from("<I", mv, 0)
    if expect is not None and n != expect:
        raise ValueError(f"batch message carries {n} arrays, expected {expect}")
    if n > 64:
        raise ValueError("malformed batch message (array count)")
    off, parts = 4, []
    for _ in range(n):
        if off + 8 > len(mv):
            raise ValueError("truncated batch message")
        (ln,) = struct.unpack_from("<Q", mv, off)
        off += 8
        if off + ln > len(mv):
            raise ValueError("truncated batch message")
        parts.append(mv[off:off + ln])
        off += ln
    if planes_of_one and n == 2:
        (c0, s0), (c1, s1) = _peek_shape(parts[0]), _peek_shape(parts[1])
        if c0 == c1 and len(s0) == 2 and len(s1) == 3 and s1[1:] == s0 and s1[0] < (1 << 16) and s0[0] * s0[1] * (s1[0] + 1) < (1 << 40):
            whole = torch.empty((s1[0] + 1,) + s0, dtype=_DTYPES[c0][0], device=device)
            return [unpack_tensor(parts[0], device, into=whole[0]), unpack_tensor(parts[1], device, into=whole[1:])]
    return [unpack_tensor(p, device) for p in parts]


def pack_public_schemes(paillier, dgk) -> bytes:
    pk = dgk.public_key
    doc = {"paillier": {"n": hex(paillier.public_key.n)},
           "dgk": {"n": hex(pk.n), "g": hex(pk.g), "h": hex(pk.h), "u": hex(pk.u), "t": pk.t,
                   "randomizer_bits": dgk.randomizer_bits, "fixed_base_window": dgk.fixed_base_window}}
    return json.dumps(doc).encode()


def unpack_public_schemes(buf: bytes, engine=None):
    from .schemes import DGK, Paillier

    raw = bytes(_as_view(buf)[:MAX_SCHEME_DOCUMENT + 1])
    if len(raw) > MAX_SCHEME_DOCUMENT:               # (five integers of the largest supported keys are ~10 KB of hex)
        raise ValueError("scheme document too large")
    try:
        doc = json.loads(raw.decode())
        d = doc["dgk"]
        int(doc["paillier"]["n"], 16), [int(d[k], 16) for k in ("n", "g", "h", "u")], int(d["t"]), int(d["randomizer_bits"]), int(d["fixed_base_window"])
    except (KeyError, TypeError, AttributeError, UnicodeDecodeError, json.JSONDecodeError) as exc:
        raise ValueError(f"malformed scheme document: {exc!r}") from None
    n_dgk, rbits = int(d["n"], 16), int(d["randomizer_bits"])
    if not 1 <= rbits <= 2 * n_dgk.bit_length():
        raise ValueError(f"scheme document asks for {rbits}-bit DGK randomizers")
    # the sender's table window is a hint about ITS memory: a peer does not get to size this party's tables (window 20 is 6 GB at
    # 2048 bits; 16 costs 1.4 % of a step and 0.7 GB).  A party that wants more pre-sets its scheme objects.
    window = max(1, min(int(d["fixed_base_window"]), MAX_WINDOW_FROM_PEER))
    return (Paillier(int(doc["paillier"]["n"], 16), engine=engine),
            DGK(n_dgk, int(d["g"], 16), int(d["h"], 16), int(d["u"], 16), int(d["t"]), engine=engine, randomizer_bits=rbits, fixed_base_window=window))


# ---- messages of the ONE-comparison protocol as bytes -----------------------------------------------------------------------------------
# The reference hands ciphertext OBJECTS to its transport, whose serializer (tno.mpc.communication + the schemes' hooks: out of scope,
# SURVEY 2.2) turns each into bytes and finds the receiver's scheme instance again.  This is the minimal equivalent for the four
# messages of SC/initiator.py:69-175 / SC/keyholder.py:80-133, so that concurrent single comparisons can run between two processes:
#
#     message = magic "SCO1" | paillier words u16 | dgk words u16 | shape length u32 | shape | rows (little-endian words, in order)
#     shape   = 'P' | 'D'                       one Paillier / DGK ciphertext
#             | 'L' count u32 shape*            a list            | 'T' count u32 shape*      a tuple
#
# The receiver binds every ciphertext to ITS scheme objects and checks the word counts against them; the rows of a list stay one
# array (coalesce.rows_of hands them to the next batched step without making integers).
OBJ_MAGIC = b"SCO1"


def pack_session_message(message: Any, paillier, dgk) -> bytes:
    """`message` -- what a transport's serializer would see, i.e. after communicator._as_on_wire -- as bytes."""
    from .coalesce import rows_of
    from .schemes import DGKCiphertext, PaillierCiphertext

    pw, dw = 2 * paillier.mod_n.nwords, dgk.mod_n.nwords
    leaf = {PaillierCiphertext: (0x50, pw), DGKCiphertext: (0x44, dw)}
    shape, rows = bytearray(), []

    def walk(m: Any) -> None:
        kind = type(m)
        if kind in leaf:
            tag, nwords = leaf[kind]
            shape.append(tag)
            rows.append(rows_of([m], nwords))
        elif kind is list or kind is tuple:
            shape.append(0x4C if kind is list else 0x54)
            shape.extend(struct.pack("<I", len(m)))
            first = type(m[0]) if m else None
            if first in leaf and all(type(c) is first for c in m):      # the l + 1 values of steps 4b / 4i, the three of step 5: one block
                tag, nwords = leaf[first]
                shape.extend(bytes([tag]) * len(m))
                rows.append(rows_of(m, nwords))
            else:
                for c in m:
                    walk(c)
        else:
            raise TypeError(f"the one-comparison protocol sends ciphertexts, lists and tuples of them, not {kind.__name__}")

    walk(message)
    shape.extend(b"\0" * (-len(shape) % 4))                             # (rows start on a word boundary)
    parts = [OBJ_MAGIC, struct.pack("<HHI", pw, dw, len(shape)), bytes(shape)]
    for r in rows:
        parts.append(np.ascontiguousarray(r.array() if hasattr(r, "array") else r, dtype="<u4").tobytes())
    out = b"".join(parts)
    STATS["bytes"] += len(out)
    return out


def unpack_session_message(buf: Any, paillier, dgk) -> Any:
    """Inverse of pack_session_message, bound to the receiver's schemes.  Nothing is trusted: word counts must be this party's, the
    shape must parse completely and the rows must fill the message exactly."""
    from .schemes import DGKCiphertext, PaillierCiphertext

    mv = _as_view(buf)
    if len(mv) < 12 or bytes(mv[:4]) != OBJ_MAGIC:
        raise ValueError("not a message of the one-comparison protocol")
    pw, dw, ns = struct.unpack("<HHI", mv[4:12])
    if pw != 2 * paillier.mod_n.nwords or dw != dgk.mod_n.nwords:
        raise ValueError(f"message carries {pw}- / {dw}-word ciphertexts, this party's schemes have {2 * paillier.mod_n.nwords} / {dgk.mod_n.nwords}")
    if ns > (1 << 20) or 12 + ns > len(mv) or ns % 4 or (len(mv) - 12 - ns) % 4:
        raise ValueError("malformed message (shape length)")
    shape = bytes(mv[12:12 + ns])
    words = np.frombuffer(mv[12 + ns:], dtype="<u4")
    leaf = {0x50: (PaillierCiphertext, paillier, pw), 0x44: (DGKCiphertext, dgk, dw)}
    pos, at = 0, 0
    # a message of one kind of ciphertext -- all four of the protocol are -- is ONE array of rows, whatever its nesting: ([d], [[beta_i]])
    # arrives as rows 0 .. l of one block, and the next batched step takes them as a slice (coalesce.rows_of)
    wholes: dict[int, np.ndarray] = {}

    def take(tag: int, count: int) -> list:
        nonlocal at
        cls, scheme, nwords = leaf[tag]
        if at + count * nwords > len(words):
            raise ValueError("malformed message (fewer rows than the shape announces)")
        if at % nwords == 0 and len(words) % nwords == 0:
            whole = wholes.get(nwords)
            if whole is None:
                whole = wholes[nwords] = words.reshape(-1, nwords)
            out = cls.rows(whole, scheme, start=at // nwords, count=count)
        else:
            out = cls.rows(words[at:at + count * nwords].reshape(count, nwords), scheme)
        at += count * nwords
        return out

    def walk(depth: int) -> Any:
        nonlocal pos
        if pos >= len(shape) or depth > 4:
            raise ValueError("malformed message (shape)")
        tag = shape[pos]
        pos += 1
        if tag in leaf:
            return take(tag, 1)[0]
        if tag in (0x4C, 0x54):
            if pos + 4 > len(shape):
                raise ValueError("malformed message (shape)")
            (count,) = struct.unpack("<I", shape[pos:pos + 4])
            pos += 4
            if count > len(shape) - pos:
                raise ValueError("malformed message (element count)")
            if count and shape[pos] in leaf and shape[pos:pos + count] == bytes([shape[pos]]) * count:
                items = take(shape[pos], count)
                pos += count
            else:
                items = [walk(depth + 1) for _ in range(count)]
            return items if tag == 0x4C else tuple(items)
        raise ValueError("malformed message (unknown shape tag)")

    out = walk(0)
    if shape[pos:].strip(b"\0") or at != len(words):
        raise ValueError("malformed message (trailing bytes)")
    return out
