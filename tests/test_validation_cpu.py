"""Everything a peer or a caller can get wrong at the edges of the batch path is refused with ValueError before a kernel
sees a pointer: malformed wire messages, arrays of the wrong dtype / shape / device, and the plaintext encodings the
reference's signature advertises (SC/initiator.py:69-72: `PaillierCiphertext | float`)."""
import asyncio
import random
import struct
import warnings

import pytest
import torch

from _comm import DictionaryCommunicator
from _oracle_engine import OracleEngine
from conftest import oracle_dgk, oracle_paillier
from oracle import sc_oracle as o
from protocols.secure_comparison_amd import DGK, Initiator, KeyHolder, Paillier, wire

L = 16


@pytest.fixture(scope="module")
def world(keys):
    osk, od = oracle_paillier(keys, 1024), oracle_dgk(keys, "dgk_tiny_l16")
    eng = OracleEngine()
    bob_p = Paillier(osk.n, osk.p, osk.q, engine=eng)
    bob_d = DGK(od.n, od.g, od.h, od.u, od.t, od.p, od.q, od.v_p, od.v_q, full_decryption=True, engine=eng, randomizer_bits=50)
    return osk, od, eng, bob_p, bob_d


def test_wire_header_is_checked_against_the_payload():
    t = torch.arange(24, dtype=torch.int32).reshape(2, 3, 4)
    buf = bytes(wire.pack_tensor(t))
    assert torch.equal(wire.unpack_tensor(buf), t)
    bad_dims = buf[:8] + struct.pack("<3Q", 2, 3, 400) + buf[32:]          # announces more items than it carries
    bad_code = buf[:4] + struct.pack("<BBH", 9, 3, 0) + buf[8:]            # unknown dtype code
    bad_ndim = buf[:4] + struct.pack("<BBH", 0, 200, 0) + buf[8:]          # absurd rank
    for bad in (bad_dims, bad_code, bad_ndim, buf[:-4], buf[:6], b""):
        with pytest.raises(ValueError):
            wire.unpack_tensor(bad)
    many = bytes(wire.pack_many(t, t))
    assert len(wire.unpack_many(many, expect=2)) == 2
    for bad in (many[:-1], many[:10], struct.pack("<I", 3) + many[4:]):
        with pytest.raises(ValueError):
            wire.unpack_many(bad)
    with pytest.raises(ValueError):
        wire.unpack_many(many, expect=3)
    with pytest.raises(ValueError):
        wire.expect_array(t, (2, 3, 5), "t")
    with pytest.raises(ValueError):
        wire.expect_array(t.to(torch.int64), (2, 3, 4), "t")


class _Tamper(DictionaryCommunicator):
    """Replaces the message with label `target` by `forge(original)` on its way to the receiver."""

    def __init__(self, box, target, forge, device_tensors=False):
        super().__init__(box, device_tensors)
        self.target, self.forge = target, forge

    async def recv(self, party_id, msg_id):
        msg = await super().recv(party_id, msg_id)
        return self.forge(msg) if msg_id.startswith(self.target) else msg


def _run_pair(world, alice_comm, bob_comm, B=3):
    osk, od, eng, bob_p, bob_d = world
    rng = random.Random(3)
    nw = bob_p.mod_n.nwords
    tx = eng.upload([osk.enc_raw(rng.randrange(1 << L)) for _ in range(B)], 2 * nw)
    ty = eng.upload([osk.enc_raw(rng.randrange(1 << L)) for _ in range(B)], 2 * nw)
    alice = Initiator(L, alice_comm, "bob")
    bob = KeyHolder(L, bob_comm, "alice", bob_p, bob_d)

    async def go():
        # a refused message leaves the other player waiting: run Alice and Bob as tasks and cancel the survivor
        ta = asyncio.ensure_future(alice.perform_secure_comparison_batch(tx, ty, None, engine=eng))
        tb = asyncio.ensure_future(bob.perform_secure_comparison_batch(None))
        done, pending = await asyncio.wait({ta, tb}, return_when=asyncio.FIRST_EXCEPTION)
        for t in pending:
            t.cancel()
        for t in done:
            t.result()
        return ta.result()

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return asyncio.run(go())


@pytest.mark.parametrize("target, forge", [
    # Bob -> Alice: [d], [beta_i].  A peer announcing another l or another batch size must not steer Alice's kernels.
    ("step_4b_batch", lambda m: wire.pack_many(*[t[..., :-1].contiguous() for t in wire.unpack_many(m)])),          # narrower residues
    ("step_4b_batch", lambda m: wire.pack_many(wire.unpack_many(m)[0], wire.unpack_many(m)[1][:-1].contiguous())),   # l - 1 planes
    ("step_4b_batch", lambda m: wire.pack_many(wire.unpack_many(m)[0], wire.unpack_many(m)[1].to(torch.uint8))),     # byte array
    ("step_4b_batch", lambda m: wire.pack_many(wire.unpack_many(m)[0][:-1].contiguous(), wire.unpack_many(m)[1])),   # B - 1 items
    ("step_4b_batch", lambda m: wire.pack_many(wire.unpack_many(m)[0])),                                             # one array missing
    ("step_5_batch", lambda m: wire.pack_many(*[t.reshape(-1) for t in wire.unpack_many(m)])),                      # flattened
    ("step_5_batch", lambda m: wire.pack_many(*[t.to(torch.int64) for t in wire.unpack_many(m)])),
])
def test_alice_refuses_malformed_batches(world, target, forge):
    box = {}
    with pytest.raises(ValueError):
        _run_pair(world, _Tamper(box, target, forge), DictionaryCommunicator(box, False))


@pytest.mark.parametrize("target, forge", [
    ("step_1_batch", lambda m: wire.pack_many(wire.unpack_many(m)[0][:, :-1].contiguous())),
    ("step_1_batch", lambda m: wire.pack_many(wire.unpack_many(m)[0].reshape(-1))),
    ("step_1_batch", lambda m: wire.pack_many(wire.unpack_many(m)[0].to(torch.uint8))),
    ("step_1_batch", lambda m: wire.pack_tensor(wire.unpack_many(m)[0])),                                              # unframed array
    ("step_4i_batch", lambda m: wire.pack_many(wire.unpack_many(m)[0][1:].contiguous())),                             # l planes instead of l + 1
    ("step_4i_batch", lambda m: wire.pack_many(wire.unpack_many(m)[0][:, :-1].contiguous())),                         # B - 1 comparisons
    ("step_4i_batch", lambda m: wire.pack_many(wire.unpack_many(m)[0].to(torch.int64))),
    ("step_4i_batch", lambda m: wire.pack_many(*wire.unpack_many(m), *wire.unpack_many(m))),                          # one array too many
])
def test_bob_refuses_malformed_batches(world, target, forge):
    box = {}
    with pytest.raises(ValueError):
        _run_pair(world, DictionaryCommunicator(box, False), _Tamper(box, target, forge))


@pytest.mark.parametrize("side, target, forge", [
    # the device hand-over form (wire.DeviceArrays) is checked like bytes: array count, dtype, shape -- and that it is arrays at all
    ("alice", "step_4b_batch", lambda m: wire.DeviceArrays((m.arrays[0],))),
    ("alice", "step_4b_batch", lambda m: wire.DeviceArrays((m.arrays[0], m.arrays[1][:-1]))),
    ("alice", "step_5_batch", lambda m: wire.DeviceArrays(tuple(t.to(torch.int64) for t in m.arrays))),
    ("alice", "step_5_batch", lambda m: wire.DeviceArrays((m.arrays[0], m.arrays[1], "zeta"))),
    ("bob", "step_1_batch", lambda m: wire.DeviceArrays((m.arrays[0][:, :-1],))),
    ("bob", "step_1_batch", lambda m: wire.DeviceArrays((None,))),
    ("bob", "step_4i_batch", lambda m: wire.DeviceArrays((m.arrays[0][:, :-1],))),
    ("bob", "step_4i_batch", lambda m: b"SCB1" + bytes(40)),
])
def test_device_hand_over_is_checked_too(world, side, target, forge):
    box = {}
    tamper, plain = _Tamper(box, target, forge, device_tensors=True), DictionaryCommunicator(box, True)
    with pytest.raises(ValueError):
        _run_pair(world, *((tamper, plain) if side == "alice" else (plain, tamper)))


def test_untampered_pair_still_runs(world):
    osk = world[0]
    box = {}
    res = _run_pair(world, DictionaryCommunicator(box), DictionaryCommunicator(box))
    assert all(osk.dec_raw(v) in (0, 1) for v in world[2].download(res))


def test_engine_array_checks_run_before_any_pointer_is_taken():
    """Engine._arr is what every C-ABI wrapper passes its arrays through; it needs no GPU to say no."""
    from protocols.secure_comparison_amd.engine import Engine

    e = object.__new__(Engine)
    e.device = torch.device("cpu")
    good = torch.zeros((4, 8), dtype=torch.int32)
    assert e._arr(good, "x", 4, 8) is good and e._arr(None, "x", optional=True) is None
    assert e._arr(torch.zeros(8, dtype=torch.int32), "x", 4, 8, broadcast=True) is not None
    cases = [(good.to(torch.uint8), 4, 8), (good.to(torch.int64), 4, 8), (good[:, ::2], 4, 4), (good, 5, 8), (good, 4, 7),
             (good.reshape(-1), 4, 8), (torch.tensor(3, dtype=torch.int32), None, None), ([1, 2, 3], None, None), (None, None, None)]
    for t, rows, words in cases:
        with pytest.raises(ValueError):
            e._arr(t, "x", rows, words)
    e.device = torch.device("meta")
    with pytest.raises(ValueError):
        e._arr(good, "x", 4, 8)


def test_plaintext_encodings(world):
    """Integral floats and fixed-point floats at the unsafe_encrypt edge (the reference takes `PaillierCiphertext | float`
    and encodes with the scheme's precision, SC/initiator.py:93-102); DGK plaintexts stay integers."""
    osk, od, eng, bob_p, bob_d = world
    assert bob_p.unsafe_encrypt(42.0).value == osk.enc_raw(42) and bob_p.decrypt(bob_p.unsafe_encrypt(-7.0)) == -7
    assert isinstance(bob_p.decrypt(bob_p.unsafe_encrypt(5)), int)
    fx = Paillier(osk.n, osk.p, osk.q, engine=eng, precision=3)
    assert fx.unsafe_encrypt(1.5).value == osk.enc_raw(1500) and fx.unsafe_encrypt(-0.001).value == osk.enc_raw(osk.n - 1)
    assert fx.unsafe_encrypt(0.1).value == osk.enc_raw(100) and fx.unsafe_encrypt(2).value == osk.enc_raw(2000)
    assert fx.decrypt(fx.unsafe_encrypt(-2.25)) == -2.25 and fx.decrypt(fx.unsafe_encrypt(3), apply_encoding=False) == 3000
    assert fx.public_copy().precision == 3
    with pytest.warns(UserWarning, match="decimal digits"):
        assert fx.unsafe_encrypt(0.12345).value == osk.enc_raw(123)
    with pytest.warns(UserWarning, match="decimal digits"):
        assert bob_p.unsafe_encrypt(2.5).value == osk.enc_raw(3)            # precision 0: ties away from zero
    for bad in (float("nan"), float("inf")):
        with pytest.raises(ValueError):
            bob_p.unsafe_encrypt(bad)
    assert bob_d.unsafe_encrypt(3.0).value == od.enc_raw(3)
    with pytest.raises(ValueError):
        bob_d.unsafe_encrypt(0.5)
    # floats through the whole single-comparison protocol (a precision-0 scheme, like the reference's default one: the protocol's
    # own `1 - [[delta_B]]` adds an ENCODED 1, so it is only meaningful when the encoding is the identity on integers)
    box = {}
    alice = Initiator(L, DictionaryCommunicator(box), "bob")
    bob = KeyHolder(L, DictionaryCommunicator(box), "alice", bob_p, bob_d)

    async def go(a, b):
        res, _ = await asyncio.gather(alice.perform_secure_comparison(a, b), bob.perform_secure_comparison())
        return res

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert bob_p.decrypt(asyncio.run(go(41.0, 42.0))) == 1 and bob_p.decrypt(asyncio.run(go(42.0, 41.0))) == 0
        assert bob_p.decrypt(asyncio.run(go(7.4, 7.0))) == 1        # 7.4 is rounded to 7 by the precision-0 encoding


def test_scalar_multiplication_registers_nothing(world):
    """ct * k sends k as data (modexp_var): the engine's exponent registry does not grow with the scalars used."""
    osk, od, eng, bob_p, bob_d = world
    calls = []
    orig = eng.modexp_shared
    eng.modexp_shared = lambda *a, **k: (calls.append(a), orig(*a, **k))[1]
    try:
        c = bob_d.unsafe_encrypt(5)
        rng = random.Random(1)
        for _ in range(5):
            k = 1 + rng.randrange(od.u - 1)
            assert (c * k).value == od.mul(c.value, k)
        assert (bob_p.unsafe_encrypt(3) * 12345).value == osk.mul(osk.enc_raw(3), 12345) and (c * 0).value == 1
    finally:
        eng.modexp_shared = orig
    assert not calls


def test_zero_copy_row_joins():
    """_views.cat_rows: a view when the parts are consecutive blocks of one buffer (what the batch steps hand around), a copy
    otherwise -- never a wrong answer."""
    from protocols.secure_comparison_amd._views import cat_rows

    a = torch.arange(5 * 3 * 4, dtype=torch.int32).reshape(5, 3, 4)
    v = cat_rows([a[0:1], a[1:]])
    assert v.data_ptr() == a.data_ptr() and torch.equal(v, a)
    v = cat_rows([a[1:2], a[2:4]])
    assert v.data_ptr() == a[1:].data_ptr() and torch.equal(v, a[1:4])
    for parts in ([a[0:1], a[2:]], [a[1:], a[0:1]], [a[:2], a[:2]], [a[:, :2].contiguous()[:1], a[:, :2].contiguous()[1:]][::-1],
                  [a[::2], a[1::2]]):
        w = cat_rows(parts)
        assert torch.equal(w, torch.cat(list(parts), dim=0))
    b = torch.arange(12, dtype=torch.int32).reshape(6, 2)
    x = cat_rows([b[:2], b[2:4], b[4:]])
    assert x.data_ptr() == b.data_ptr() and torch.equal(x, b)
    assert cat_rows([b[:3], b[3:].to(torch.int64)]).dtype == torch.int64      # mixed dtypes: plain torch.cat semantics


def test_kernel_written_message_layout_and_padding(monkeypatch):
    """wire.Reserved lays a byte message out up front (payloads on 256-byte boundaries of the buffer, the padding announced in the
    array header) so that the producing kernels can write it in place; the receiver parses it like a packed message, and a header
    whose padding does not fit the bytes that arrived is refused.  (On the GPU the buffer is pinned memory from the pool; here a
    plain one stands in -- the layout code is the same.)"""
    import struct

    from protocols.secure_comparison_amd import wire

    class PlainPool:
        def take(self, nbytes):
            buf = torch.zeros(nbytes + 64, dtype=torch.uint8)[:nbytes]
            return buf, buf.numpy()

    monkeypatch.setattr(wire, "_pinned_pool", PlainPool())
    shapes = [(5, 7), (3, 5, 7), (0, 4)]
    msg = wire.Reserved("cpu", shapes)
    for i, a in enumerate(msg.arrays):
        assert tuple(a.shape) == shapes[i] and a.dtype == torch.int32
        assert (a.data_ptr() - msg._buf.data_ptr()) % wire.Reserved.ALIGN == 0 or a.numel() == 0
        a.copy_(torch.arange(a.numel(), dtype=torch.int32).reshape(shapes[i]) + 1000 * i)
    raw = bytes(memoryview(msg._raw))
    got = wire.unpack_many(raw, "cpu", expect=3)
    assert [tuple(t.shape) for t in got] == shapes and got[1][2, 4, 6].item() == 1000 + 3 * 5 * 7 - 1
    joined = wire.unpack_many(bytes(memoryview(wire.Reserved("cpu", [(5, 7), (3, 5, 7)])._raw)), "cpu", expect=2, planes_of_one=True)
    assert joined[1].data_ptr() == joined[0].data_ptr() + 5 * 7 * 4          # [d] and the planes [beta_i] as one array
    # a padding that runs past the message, or an absurd one, is refused
    one = wire.Reserved("cpu", [(2, 2)])
    mv = bytearray(memoryview(one._raw))
    (ln,) = struct.unpack_from("<Q", mv, 4)
    struct.pack_into("<H", mv, 4 + 8 + 6, 5000)
    with pytest.raises(ValueError):
        wire.unpack_many(bytes(mv), "cpu")
    struct.pack_into("<H", mv, 4 + 8 + 6, 3)                                   # another padding: the payload size no longer matches
    with pytest.raises(ValueError):
        wire.unpack_many(bytes(mv), "cpu")


def test_forged_plane_header_is_refused_before_anything_is_allocated():
    """Round-4 advice: with planes_of_one the receiver sized ONE array for [d] and the planes [beta_i] from the two announced shapes
    before either payload had been compared with its header -- a 76-byte message announcing [2^20][2^10] and [255][2^20][2^10] made
    it try a terabyte allocation (RuntimeError, or a real 200 GB array on the GPU).  The announced shape is now held against the
    bytes that arrived first: ValueError, in both forms, and a well-formed pair still joins."""
    import struct

    from protocols.secure_comparison_amd import wire

    def array_msg(shape, payload=b""):
        return wire.MAGIC + struct.pack("<BBH", 0, len(shape), 0) + struct.pack(f"<{len(shape)}Q", *shape) + payload

    a, b = array_msg((1 << 20, 1 << 10)), array_msg((255, 1 << 20, 1 << 10))
    forged = struct.pack("<I", 2) + struct.pack("<Q", len(a)) + a + struct.pack("<Q", len(b)) + b
    assert len(forged) < 100
    for planes in (True, False):
        with pytest.raises(ValueError, match="announces shape"):
            wire.unpack_many(forged, "cpu", planes_of_one=planes)
    # only the second header lies: still refused before the joint array exists
    d = torch.arange(6, dtype=torch.int32).reshape(2, 3)
    good_a = bytes(wire.pack_tensor(d))
    lying = struct.pack("<I", 2) + struct.pack("<Q", len(good_a)) + good_a + struct.pack("<Q", len(array_msg((60000, 2, 3)))) + array_msg((60000, 2, 3))
    with pytest.raises(ValueError, match="announces shape"):
        wire.unpack_many(lying, "cpu", planes_of_one=True)
    planes_t = torch.arange(24, dtype=torch.int32).reshape(4, 2, 3)
    got = wire.unpack_many(bytes(wire.pack_many(d, planes_t)), "cpu", expect=2, planes_of_one=True)
    assert torch.equal(got[0], d) and torch.equal(got[1], planes_t) and got[1].data_ptr() == got[0].data_ptr() + 24


def test_scheme_documents_from_a_peer_are_bounded(keys):
    """wire.unpack_public_schemes: what arrives is a small JSON document of hex integers; anything else is a ValueError before a scheme
    object exists, and the sender's fixed-base window -- a statement about ITS memory -- cannot size this party's tables."""
    import json

    od, osk = oracle_dgk(keys, "dgk_tiny_l16"), oracle_paillier(keys, 1024)
    eng = OracleEngine()
    good = wire.pack_public_schemes(Paillier(osk.n, engine=eng), DGK(od.n, od.g, od.h, od.u, od.t, engine=eng, randomizer_bits=50, fixed_base_window=20))
    pai, dgk = wire.unpack_public_schemes(good, eng)
    assert pai.public_key.n == osk.n and dgk.public_key.h == od.h and dgk.randomizer_bits == 50
    assert dgk.fixed_base_window == wire.MAX_WINDOW_FROM_PEER == 16
    doc = json.loads(good)
    for mutate in (lambda d: d["dgk"].pop("h"), lambda d: d["dgk"].update(n="zz"), lambda d: d["dgk"].update(randomizer_bits=10 ** 9),
                   lambda d: d["dgk"].update(randomizer_bits=0), lambda d: d.pop("paillier"), lambda d: d["dgk"].update(fixed_base_window=None),
                   lambda d: d["paillier"].update(n=7)):
        bad = json.loads(json.dumps(doc))
        mutate(bad)
        with pytest.raises(ValueError):
            wire.unpack_public_schemes(json.dumps(bad).encode(), eng)
    for raw in (b"", b"[1, 2]", b"\xff\xfe", good + b" " * (1 << 16)):
        with pytest.raises(ValueError):
            wire.unpack_public_schemes(raw, eng)
