"""Concurrent single comparisons as batch launches.

The reference's call shape is ONE comparison per `perform_secure_comparison` call, concurrency by asyncio sessions of one
Initiator / KeyHolder pair whose messages are kept apart by the session number
(/root/reference/src/tno/mpc/protocols/secure_comparison/initiator.py:69-175, :86-87; test/unit/test_secure_comparison.py:804-835).
On the GPU a lone comparison is a chain of dependent launches on an otherwise idle chip (8.9 ms): N concurrent sessions
that each walk that chain cost N times as much, although the chip could take thousands of them in the same launches.

`StepCoalescer` closes that gap without changing the call shape: a session hands each of its steps to the coalescer of its player
object and awaits the result; requests of the same step that arrive within the same turn of the event loop (plus an optional
linger, or until `max_batch` are waiting) are executed as ONE call of the batch entry points (sc_initiator_step1 / _step4 /
_step67, sc_keyholder_step2_4b / _step4j_5) and the rows handed back to their sessions.  Every session still makes its own random
draws, in its own task, in the order the single path makes them -- so its messages are the same integers as in an uncoalesced run
with the same draws (tests/test_gpu_round5.py, tests/test_coalesce_cpu.py).
"""
from __future__ import annotations

import asyncio
import contextlib
import gc
import threading
import time
from typing import Any, Callable, Sequence

import numpy as np

from .limbs import RowBlock


class _CollectorPause:
    """The cyclic garbage collector, paused for the start of a burst of sessions -- process-wide state, so kept in ONE place for every
    coalescer of the process (the two players of a test or a bench share an interpreter).

    A thousand sessions in flight keep ~10^5 small container objects alive (tasks, futures, messages of l + 1 ciphertext objects).
    With CPython's default thresholds the collector examines them -- and, in its full passes, everything torch and numpy imported --
    a few hundred times per burst, and finds nothing: the objects die by reference count when their sessions end.  Measured on the
    GPU box: 5.8 k -> 7.9 k comparisons/s at l = 32 / 2048-bit, 7.8 k -> 12.9 k at l = 16 / 1024-bit.  So the collector is switched
    off when the first session of a burst enters and on again when the last one leaves -- or `limit_s` after the first one entered,
    whichever comes first: a server whose sessions never drain loses nothing but the first `limit_s` of collection per burst, and
    cyclic garbage made meanwhile is collected right afterwards.  Nothing happens when the application has the collector off
    already, and `pause_collector_s = 0` on a player keeps the library's hands off it."""

    def __init__(self) -> None:
        self._lock = threading.Lock()
        self.holders = 0            # sessions in flight, over every coalescer that asked for the pause
        self.owned = False          # the collector is off because of us
        self.expired = False        # this burst has used up its limit: not again before every session has left
        self.since = self.limit = 0.0

    def enter(self, limit_s: float) -> None:
        with self._lock:
            self.holders += 1
            if self.holders == 1:
                self.expired = False
            if limit_s > 0 and not self.owned and not self.expired and gc.isenabled():
                gc.disable()
                self.owned, self.since, self.limit = True, time.monotonic(), limit_s

    def leave(self) -> None:
        with self._lock:
            self.holders -= 1
            if self.owned and self.holders <= 0:
                gc.enable()
                self.owned = False

    def check(self) -> None:
        """Called from inside a burst (after every batched call): give the collector back once the burst has had its share."""
        if self.owned and time.monotonic() - self.since > self.limit:
            with self._lock:
                if self.owned:
                    gc.enable()
                    self.owned, self.expired = False, True


_collector_pause = _CollectorPause()


class _Queue:
    __slots__ = ("items", "futures", "run", "armed", "seen", "lingered")

    def __init__(self) -> None:
        self.items: list = []
        self.futures: list[asyncio.Future] = []
        self.run: Callable[[list], Sequence] | None = None
        self.armed = False
        self.seen = 0
        self.lingered = False


class StepCoalescer:
    """Gathers the step requests of the concurrent sessions of ONE player object.

    max_batch: a queue that reaches this many requests is executed at once.  linger_s: after a turn of the event loop that brought
    no new request, wait this much longer for stragglers -- again after every straggler, so a queue is executed once nothing has
    arrived for linger_s (0: at the first quiet turn; a transport between two processes or hosts delivers the messages of concurrent
    sessions a socket read apart -- microseconds to milliseconds -- and a batch cut in two there stays cut for the rest of the
    protocol: tools/gpu_two_process_sessions.py).  The batched call itself runs synchronously on the event loop's thread, like every
    GPU call of the single path."""

    def __init__(self, max_batch: int = 4096, linger_s: float = 0.0, pause_collector_s: float = 0.5) -> None:
        self.max_batch = max(1, int(max_batch))
        self.linger_s = float(linger_s)
        self.pause_collector_s = float(pause_collector_s)     # _CollectorPause: 0 = never touch the garbage collector
        self._queues: dict[str, _Queue] = {}
        self.active = 0          # sessions of the player in flight (session())
        self.stats = {"calls": 0, "items": 0, "largest": 0, "fallbacks": 0, "seconds": {}}     # seconds: wall clock inside the batched calls, per step

    @contextlib.contextmanager
    def session(self):
        """Brackets one session of the player: the coalescer then knows how many sessions can still arrive at a step."""
        self.active += 1
        _collector_pause.enter(self.pause_collector_s)
        try:
            yield
        finally:
            self.active -= 1
            _collector_pause.leave()

    async def submit(self, kind: str, run: Callable[[list], Sequence], item: Any, first: bool = False) -> Any:
        """Queue `item` for the step `kind`; `run(items)` makes ONE batched call for a list of such items and returns one result per
        item, in order.  Returns this item's result (or raises what its own execution raised).  The queue is executed at once when
        every session in flight is waiting in it (nobody else can arrive: a lone session pays no extra turn of the event loop, a burst
        that moves in lock step goes the moment its last session arrives) -- except for a session's `first` request, when sessions
        created together may not all have started yet; otherwise after the first turn of the loop that brings no new request."""
        loop = asyncio.get_running_loop()
        q = self._queues.get(kind)
        if q is None:
            q = self._queues[kind] = _Queue()
        if q.futures and q.futures[0].get_loop() is not loop:       # a player object moved to another event loop with requests pending
            raise RuntimeError("a step coalescer serves one event loop at a time")
        fut = loop.create_future()
        q.items.append(item)
        q.futures.append(fut)
        q.run = run
        if len(q.items) >= self.max_batch:
            self._flush(kind)
        elif not first and 0 < self.active <= len(q.items):
            # (as a callback of its own, not inside this session's task: the session would otherwise run ahead of the others -- its
            # future is done before it awaits it -- and reach the peer first, and the peer's batch would no longer be in array order)
            q.armed = True
            loop.call_soon(self._flush, kind)
        elif not q.armed:
            q.armed, q.seen, q.lingered = True, 0, False
            loop.call_soon(self._tick, kind, loop)
        return await fut

    def _tick(self, kind: str, loop: asyncio.AbstractEventLoop) -> None:
        q = self._queues.get(kind)
        if q is None or not q.armed:
            return
        if len(q.items) != q.seen:                 # the queue grew during the last turn: sessions are still arriving
            q.seen, q.lingered = len(q.items), False           # (and the wait for stragglers starts again after the last arrival)
            loop.call_soon(self._tick, kind, loop)
        elif self.linger_s > 0 and not q.lingered:
            q.lingered = True
            loop.call_later(self.linger_s, self._tick, kind, loop)
        else:
            self._flush(kind)

    def _flush(self, kind: str) -> None:
        q = self._queues.pop(kind, None)
        if q is None or not q.items:
            return
        items, futures, run = q.items, q.futures, q.run
        self.stats["calls"] += 1
        self.stats["items"] += len(items)
        self.stats["largest"] = max(self.stats["largest"], len(items))
        t0 = time.perf_counter()
        # the batched call hands back tens of thousands of small objects (one ciphertext per row): with the cyclic collector running,
        # every few hundred allocations trigger a pass over them (measured: 0.18 s instead of 0.03 s for 1024 sessions at l = 32)
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            results = list(run(items))
            if len(results) != len(items):
                raise RuntimeError(f"batched step {kind!r} returned {len(results)} results for {len(items)} requests")
            outcomes = [(True, r) for r in results]
        except Exception as exc:  # noqa: BLE001 -- handed to the session(s) it belongs to
            if len(items) == 1:
                outcomes = [(False, exc)]
            else:
                # one session's bad input (a non-invertible ciphertext, a malformed value) must not fail its neighbours: every request
                # on its own, each session gets its own outcome
                self.stats["fallbacks"] += 1
                outcomes = []
                for it in items:
                    try:
                        outcomes.append((True, list(run([it]))[0]))
                    except Exception as one:  # noqa: BLE001
                        outcomes.append((False, one))
        finally:
            if gc_was_on:
                gc.enable()
        self.stats["seconds"][kind] = self.stats["seconds"].get(kind, 0.0) + time.perf_counter() - t0
        _collector_pause.check()
        for fut, (ok, value) in zip(futures, outcomes):
            if fut.done():                          # the session was cancelled while it waited
                continue
            if ok:
                fut.set_result(value)
            else:
                fut.set_exception(value)


# ---- word rows: how coalesced sessions hand ciphertext values on without ever making Python integers of them ------------------------
def rows_of(cts: Sequence, nwords: int):
    """The values of a list of ciphertexts as rows of little-endian 32-bit words: the very block a coalesced peer's batch launch
    produced when the ciphertexts are ALL its rows in order (a limbs.RowBlock or an array: no conversion, no copy), consecutive
    rows of one array as a slice of it, else an array [len][nwords] converted from their integers."""
    block = cts[0]._block if cts else None
    if block is not None:
        at = cts[0]._row
        for c in cts:
            if c._block is not block or c._row != at:
                break
            at += 1
        else:
            if type(block) is RowBlock:
                if cts[0]._row == 0 and at == len(block) and block.words == nwords:
                    return block
            elif block.shape[-1] == nwords:
                return block[cts[0]._row:at]
    out = np.empty((len(cts), nwords), dtype="<u4")
    for i, c in enumerate(cts):
        out[i] = np.frombuffer(c.peek_value().to_bytes(4 * nwords, "little"), dtype="<u4")
    return out


STACK_STATS = {"taken_whole": 0, "assembled": 0}      # how stack_blocks got its arrays (tests, tools)


def stack_blocks(blocks: Sequence, rows: int, words: int) -> np.ndarray:
    """The bit-major array [rows][K][words] with session i's block at [:, i]: the peer's own batch array when the K blocks are that
    array's K sessions in order (the common case: both players coalesce the same sessions), else assembled block by block."""
    first = blocks[0]
    if type(first) is RowBlock and first.b == 0 and first.base.shape == (rows, len(blocks), words):
        base = first.base
        for i, blk in enumerate(blocks):
            if type(blk) is not RowBlock or blk.base is not base or blk.b != i:
                break
        else:
            STACK_STATS["taken_whole"] += 1
            return base
    STACK_STATS["assembled"] += 1
    out = np.empty((rows, len(blocks), words), dtype="<u4")
    for i, blk in enumerate(blocks):
        out[:, i] = blk.array() if type(blk) is RowBlock else blk
    return out


def int_rows(values: Sequence[int], nwords: int) -> np.ndarray:
    """Python integers as rows of words."""
    return np.frombuffer(b"".join(v.to_bytes(4 * nwords, "little") for v in values), dtype="<u4").reshape(len(values), nwords)


@contextlib.contextmanager
def quiet_collector(generation0: int = 1_000_000):
    """For applications that serve bursts of thousands of concurrent sessions: inside this context the cyclic garbage collector's
    generation-0 threshold is raised (a collection per `generation0` container allocations instead of per 700) and the objects that
    exist already are frozen out of its passes; both are restored on exit.  A thousand sessions in flight keep ~10^5 small container
    objects alive (messages of l + 1 ciphertext objects each); CPython's default thresholds then walk them -- and everything torch
    and numpy imported -- a few hundred times per burst.  The library's own default is narrower (_CollectorPause above: the collector off
    for the start of a burst, half a second at most, nothing frozen, no threshold changed) and gets within 4 % of this; freezing the
    interpreter's objects and changing thresholds for as long as the application likes is the application's decision."""
    old = gc.get_threshold()
    gc.collect()
    gc.freeze()
    gc.set_threshold(generation0, old[1], old[2])
    try:
        yield
    finally:
        gc.set_threshold(*old)
        gc.unfreeze()
