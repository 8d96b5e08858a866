"""The batched interactive protocol between TWO PROCESSES (the reference's integration test runs its two players over an HTTP
pool pair, SC/test/integration/test_pool.py:41-73): Alice in this process, the key holder in a child process, every message a
byte string over a multiprocessing pipe -- so nothing but the documented wire format (wire.py) crosses, and the draws of each
party come from its own generator.  Arithmetic by the test-only OracleEngine (no GPU in this tier)."""
import asyncio
import multiprocessing as mp
import os
import random
import sys

import pytest

from conftest import ROOT, oracle_dgk, oracle_paillier

L = 16


class PipeCommunicator:
    """Bytes only: (msg_id, payload) frames over a duplex pipe; messages that arrive early wait in a local mailbox."""

    device_tensors = False

    def __init__(self, conn):
        self.conn, self.box = conn, {}

    async def send(self, party_id, message, msg_id):
        payload = bytes(message)                    # a memoryview from wire.pack_many, or the JSON scheme document
        self.conn.send_bytes(msg_id.encode() + b"\0" + payload)

    async def recv(self, party_id, msg_id):
        while msg_id not in self.box:
            if not self.conn.poll(60):
                raise TimeoutError(msg_id)
            frame = self.conn.recv_bytes()
            key, _, payload = frame.partition(b"\0")
            self.box[key.decode()] = payload
        return self.box.pop(msg_id)


def _keyholder_process(conn, root):
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import json

    from _oracle_engine import OracleEngine
    from conftest import oracle_dgk as od_, oracle_paillier as op_
    from protocols.secure_comparison_amd import DGK, KeyHolder, Paillier

    keys = json.load(open(os.path.join(root, "tests", "golden", "keys.json")))
    osk, od = op_(keys, 1024), od_(keys, "dgk_tiny_l16")
    eng = OracleEngine()
    bob = KeyHolder(L, PipeCommunicator(conn), "alice", Paillier(osk.n, osk.p, osk.q, engine=eng),
                    DGK(od.n, od.g, od.h, od.u, od.t, od.p, od.q, od.v_p, od.v_q, engine=eng, randomizer_bits=50))
    for _ in range(3):
        asyncio.run(bob.perform_secure_comparison_batch())
    conn.close()


def test_two_processes_bytes_only(keys):
    from _oracle_engine import OracleEngine
    from protocols.secure_comparison_amd import Initiator

    osk, od = oracle_paillier(keys, 1024), oracle_dgk(keys, "dgk_tiny_l16")
    ctx = mp.get_context("spawn")
    here, there = ctx.Pipe(duplex=True)
    child = ctx.Process(target=_keyholder_process, args=(there, ROOT), daemon=True)
    child.start()
    try:
        eng = OracleEngine()
        alice = Initiator(L, PipeCommunicator(here), "bob")
        rng = random.Random(17)
        nw = 32
        # three sessions on one connection: the session-numbered labels keep them apart; the last one travels as three chunks
        # (plan message + `.._chunk_i` sub-sessions whose messages interleave on the pipe)
        for B, chunks in ((6, 1), (1, 1), (7, 3)):
            xs = [rng.randrange(1 << L) for _ in range(B)]
            ys = [xs[i] if i % 3 == 0 else rng.randrange(1 << L) for i in range(B)]
            tx = eng.upload([osk.randomize(osk.enc_raw(x), 1 + rng.randrange(osk.n - 1)) for x in xs], 2 * nw)
            ty = eng.upload([osk.randomize(osk.enc_raw(y), 1 + rng.randrange(osk.n - 1)) for y in ys], 2 * nw)
            res = asyncio.run(alice.perform_secure_comparison_batch(tx, ty, engine=eng, chunks=chunks))
            assert [osk.dec_raw(v) for v in eng.download(res)] == [int(x <= y) for x, y in zip(xs, ys)]
        assert alice.session_id == 3 and alice.scheme_paillier.public_key.n == osk.n and alice.scheme_paillier.secret_key is None
    finally:
        child.join(30)
        if child.is_alive():
            child.kill()
    assert child.exitcode == 0


def _keyholder_over_a_socket(path, root, batches):
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import json

    from _oracle_engine import OracleEngine
    from conftest import oracle_dgk as od_, oracle_paillier as op_
    from protocols.secure_comparison_amd import DGK, KeyHolder, Paillier, StreamCommunicator

    keys = json.load(open(os.path.join(root, "tests", "golden", "keys.json")))
    osk, od = op_(keys, 1024), od_(keys, "dgk_tiny_l16")
    eng = OracleEngine()

    async def serve():
        comm = await asyncio.wait_for(StreamCommunicator.accept_unix(path), 120)
        bob = KeyHolder(L, comm, "alice", Paillier(osk.n, osk.p, osk.q, engine=eng),
                        DGK(od.n, od.g, od.h, od.u, od.t, od.p, od.q, od.v_p, od.v_q, engine=eng, randomizer_bits=50))
        for _ in range(batches):
            await bob.perform_secure_comparison_batch()
        await comm.close()

    asyncio.run(serve())


def test_two_processes_over_a_socket(keys, tmp_path):
    """The same exchange over communicator.StreamCommunicator (a Unix socket driven by the event loop): the transport the GPU-box tool
    tools/gpu_two_process.py uses.  Chunked sub-sessions interleave their frames on the one connection."""
    from _oracle_engine import OracleEngine
    from protocols.secure_comparison_amd import Initiator, StreamCommunicator

    osk, od = oracle_paillier(keys, 1024), oracle_dgk(keys, "dgk_tiny_l16")
    path = str(tmp_path / "sc.sock")
    ctx = mp.get_context("spawn")
    child = ctx.Process(target=_keyholder_over_a_socket, args=(path, ROOT, 2), daemon=True)
    child.start()
    eng = OracleEngine()
    rng = random.Random(23)

    async def go():
        comm = await StreamCommunicator.open_unix(path)            # (waits for the child to listen)
        alice = Initiator(L, comm, "bob")
        out = []
        for B, chunks in ((5, 1), (9, 3)):
            xs = [rng.randrange(1 << L) for _ in range(B)]
            ys = [xs[i] if i % 4 == 0 else rng.randrange(1 << L) for i in range(B)]
            tx, ty = eng.upload([osk.enc_raw(x) for x in xs], 64), eng.upload([osk.enc_raw(y) for y in ys], 64)
            res = await alice.perform_secure_comparison_batch(tx, ty, engine=eng, chunks=chunks)
            out.append(([osk.dec_raw(v) for v in eng.download(res)], [int(x <= y) for x, y in zip(xs, ys)]))
        with pytest.raises(TypeError):
            await comm.send("bob", object(), "not bytes")
        await comm.close()
        return out

    try:
        for got, want in asyncio.run(go()):
            assert got == want
    finally:
        child.join(60)
        if child.is_alive():
            child.kill()
    assert child.exitcode == 0


def test_stream_frames_are_bounded_before_anything_is_allocated():
    """A peer decides what the frame headers say: a payload beyond max_payload, a pile of messages nobody asked for beyond max_pending,
    or a second message under an unclaimed id end the connection with the reason -- before a buffer of the announced size exists."""
    import socket
    import struct
    import threading

    from protocols.secure_comparison_amd import StreamCommunicator

    def frame(msg_id, payload, announce=None):
        ident = msg_id.encode()
        return struct.pack("<I", len(ident)) + ident + struct.pack("<Q", len(payload) if announce is None else announce) + payload

    async def case(frames, ask, **limits):
        a, b = socket.socketpair()
        sizes = []
        comm = StreamCommunicator(a, alloc=lambda n: sizes.append(n) or bytearray(n), timeout_s=5, **limits)
        feeder = threading.Thread(target=lambda: b.sendall(b"".join(frames)))      # (a megabyte does not fit the socket's buffer)
        feeder.start()
        try:
            return bytes(await comm.recv("peer", ask)), sizes
        finally:
            await comm.close()
            b.close()
            feeder.join(10)

    got, sizes = asyncio.run(case([frame("m1", b"abc"), frame("m2", b"defg")], "m2", max_payload=8, max_pending=8))
    assert got == b"defg" and sizes == []                     # small payloads are parsed out of the read buffer
    big = bytes(range(256)) * 4100                            # from a megabyte on: received straight into a buffer from `alloc`
    got, sizes = asyncio.run(case([frame("m1", b"abc"), frame("big", big), frame("m3", b"xyz")], "big"))
    assert got == big and sizes == [len(big)]
    with pytest.raises(ConnectionError, match="max_payload"):
        asyncio.run(case([frame("big", b"", announce=1 << 50)], "big", max_payload=1 << 20))
    with pytest.raises(ConnectionError, match="max_pending"):
        asyncio.run(case([frame(f"u{i}", b"x" * 6) for i in range(3)], "never", max_payload=8, max_pending=16))
    with pytest.raises(ConnectionError, match="second message"):
        asyncio.run(case([frame("dup", b"1"), frame("dup", b"2")], "other"))
    with pytest.raises(ConnectionError, match="message id length"):
        asyncio.run(case([struct.pack("<I", 1 << 30)], "x"))


def _keyholder_sessions_over_a_socket(path, root, sessions, rounds):
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import json

    from _oracle_engine import OracleEngine
    from conftest import oracle_dgk as od_, oracle_paillier as op_
    from protocols.secure_comparison_amd import DGK, KeyHolder, Paillier, StreamCommunicator

    keys = json.load(open(os.path.join(root, "tests", "golden", "keys.json")))
    osk, od = op_(keys, 1024), od_(keys, "dgk_tiny_l16")
    eng = OracleEngine()

    async def serve():
        comm = await asyncio.wait_for(StreamCommunicator.accept_unix(path, engine=eng, timeout_s=120), 120)
        bob = KeyHolder(L, comm, "alice", Paillier(osk.n, osk.p, osk.q, engine=eng),
                        DGK(od.n, od.g, od.h, od.u, od.t, od.p, od.q, od.v_p, od.v_q, engine=eng, randomizer_bits=50))
        bob.coalesce_linger_s = 0.002                 # messages of concurrent sessions arrive a socket read apart
        for _ in range(rounds):
            await asyncio.gather(*(bob.perform_secure_comparison() for _ in range(sessions)))
        stats = bob._coalescer().stats
        assert stats["items"] == 3 * sessions * rounds and stats["fallbacks"] == 0, stats      # (his randomizers, steps 2-4b, steps 4j-5)
        assert stats["calls"] < stats["items"], stats              # sessions really shared launches
        await comm.close()

    asyncio.run(serve())


def test_concurrent_single_comparisons_between_two_processes(keys, tmp_path):
    """The reference's call shape between two processes: N concurrent `perform_secure_comparison` sessions on one Initiator here and
    one KeyHolder in a child process, over communicator.StreamCommunicator -- the scheme pair as its public document, every
    ciphertext message in wire.pack_session_message's form, each process coalescing its own sessions' steps.  Plaintext and
    encrypted inputs; every result decrypts to [x <= y]; Alice never holds a secret key."""
    from _oracle_engine import OracleEngine
    from protocols.secure_comparison_amd import Initiator, Paillier, PaillierCiphertext, StreamCommunicator

    osk = oracle_paillier(keys, 1024)
    path = str(tmp_path / "sc.sock")
    sessions, rounds = 12, 2
    ctx = mp.get_context("spawn")
    child = ctx.Process(target=_keyholder_sessions_over_a_socket, args=(path, ROOT, sessions, rounds), daemon=True)
    child.start()
    eng = OracleEngine()
    rng = random.Random(29)

    async def go():
        comm = await StreamCommunicator.open_unix(path, engine=eng, timeout_s=120)
        pub = Paillier(osk.n, engine=eng)
        alice = Initiator(L, comm, "bob", pub)
        alice.coalesce_linger_s = 0.002
        out = []
        for _ in range(rounds):
            xs = [rng.randrange(1 << L) for _ in range(sessions)]
            ys = [xs[i] if i % 4 == 0 else rng.randrange(1 << L) for i in range(sessions)]
            enc = lambda v: PaillierCiphertext(osk.randomize(osk.enc_raw(v), 1 + rng.randrange(osk.n - 1)), pub)  # noqa: E731
            ins = [(x, y) if i % 2 else (enc(x), enc(y)) for i, (x, y) in enumerate(zip(xs, ys))]
            res = await asyncio.wait_for(asyncio.gather(*(alice.perform_secure_comparison(a, b) for a, b in ins)), 300)
            out.append(([osk.dec_raw(r.peek_value()) for r in res], [int(x <= y) for x, y in zip(xs, ys)]))
        stats = alice._coalescer().stats
        await comm.close()
        return out, stats, alice

    try:
        out, stats, alice = asyncio.run(go())
        for got, want in out:
            assert got == want
        assert stats["items"] == 3 * sessions * rounds and stats["calls"] < stats["items"] and stats["fallbacks"] == 0, stats
        assert alice.scheme_dgk.secret_key is None and alice.scheme_paillier.secret_key is None
    finally:
        child.join(60)
        if child.is_alive():
            child.kill()
    assert child.exitcode == 0


def test_session_messages_as_bytes_are_checked(keys):
    """wire.pack_session_message / unpack_session_message: the four message shapes of the one-comparison protocol survive the round
    trip bound to the receiver's schemes (lists stay one block of rows), and nothing malformed gets through."""
    import struct

    from _oracle_engine import OracleEngine
    from protocols.secure_comparison_amd import DGK, DGKCiphertext, Paillier, PaillierCiphertext, wire

    osk, od = oracle_paillier(keys, 1024), oracle_dgk(keys, "dgk_tiny_l16")
    eng = OracleEngine()
    pai, dgk = Paillier(osk.n, engine=eng), DGK(od.n, od.g, od.h, od.u, od.t, engine=eng, randomizer_bits=50)
    pai2, dgk2 = Paillier(osk.n, engine=eng), DGK(od.n, od.g, od.h, od.u, od.t, engine=eng, randomizer_bits=50)
    rng = random.Random(3)
    P = lambda: PaillierCiphertext(rng.randrange(osk.n * osk.n), pai)  # noqa: E731
    D = lambda: DGKCiphertext(rng.randrange(od.n), dgk)  # noqa: E731
    for msg in (P(), (D(), [D() for _ in range(L)]), [D() for _ in range(L + 1)], (P(), P(), P()), [], (P(), [D(), P()])):
        back = wire.unpack_session_message(wire.pack_session_message(msg, pai, dgk), pai2, dgk2)

        def same(a, b):
            if isinstance(a, (list, tuple)):
                return type(a) is type(b) and len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
            return type(a) is type(b) and a.peek_value() == b.peek_value() and b.scheme is (pai2 if type(a) is PaillierCiphertext else dgk2) and not b.fresh

        assert same(msg, back)
    lst = wire.unpack_session_message(wire.pack_session_message([D() for _ in range(5)], pai, dgk), pai2, dgk2)
    assert all(c.peek_words()[0] is lst[0].peek_words()[0] and c.peek_words()[1] == i for i, c in enumerate(lst))
    good = wire.pack_session_message((D(), [D(), D()]), pai, dgk)
    for bad in (good[:-4], good + b"\0\0\0\0", b"XXXX" + good[4:], good[:4] + struct.pack("<HH", 7, 7) + good[8:],
                good[:8] + struct.pack("<I", 1 << 30) + good[12:], good[:12] + b"Q" + good[13:],
                good[:13] + struct.pack("<I", 1 << 31) + good[17:], good[:5]):
        with pytest.raises(ValueError):
            wire.unpack_session_message(bad, pai2, dgk2)
    with pytest.raises(TypeError):
        wire.pack_session_message([1, 2], pai, dgk)
    other = Paillier(oracle_paillier(keys, 2048).n, engine=eng)
    with pytest.raises(ValueError, match="word"):
        wire.unpack_session_message(good, other, dgk2)


def test_mutated_session_messages_never_escape_as_anything_but_valueerror(keys):
    """Two thousand random mutations (bit flips, truncations, insertions, spliced headers) of valid one-comparison messages: each either
    parses into ciphertexts of this party's schemes or raises ValueError -- no other exception, no object bound to anything else."""
    from _oracle_engine import OracleEngine
    from protocols.secure_comparison_amd import DGK, DGKCiphertext, Paillier, PaillierCiphertext, wire

    osk, od = oracle_paillier(keys, 1024), oracle_dgk(keys, "dgk_tiny_l16")
    eng = OracleEngine()
    pai, dgk = Paillier(osk.n, engine=eng), DGK(od.n, od.g, od.h, od.u, od.t, engine=eng, randomizer_bits=50)
    rng = random.Random(77)
    P = lambda: PaillierCiphertext(rng.randrange(osk.n * osk.n), pai)  # noqa: E731
    D = lambda: DGKCiphertext(rng.randrange(od.n), dgk)  # noqa: E731
    seeds = [wire.pack_session_message(m, pai, dgk) for m in (P(), (D(), [D() for _ in range(L)]), [D() for _ in range(L + 1)], (P(), P(), P()))]

    def leaves(m):
        return [x for c in m for x in leaves(c)] if isinstance(m, (list, tuple)) else [m]

    parsed = refused = 0
    for _ in range(2000):
        b = bytearray(rng.choice(seeds))
        for _ in range(rng.randrange(1, 4)):
            kind = rng.randrange(5)
            if kind == 0 and b:
                b[rng.randrange(min(len(b), 40))] ^= 1 << rng.randrange(8)           # the header and shape bytes
            elif kind == 1 and b:
                b[rng.randrange(len(b))] = rng.randrange(256)
            elif kind == 2:
                del b[rng.randrange(len(b) + 1):]
            elif kind == 3:
                at = rng.randrange(len(b) + 1)
                b[at:at] = bytes(rng.randrange(256) for _ in range(rng.choice([1, 4, 128])))
            else:
                other = rng.choice(seeds)
                b[:rng.randrange(8, 24)] = other[:rng.randrange(8, 24)]
        try:
            out = wire.unpack_session_message(bytes(b), pai, dgk)
        except ValueError:
            refused += 1
            continue
        parsed += 1
        for c in leaves(out):
            assert (type(c) is PaillierCiphertext and c.scheme is pai) or (type(c) is DGKCiphertext and c.scheme is dgk)
            c.peek_value()
    assert parsed > 50 and refused > 500, (parsed, refused)
