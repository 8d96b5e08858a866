"""The cipher core of the device-side generator, pinned on the CPU: oracle/chacha_rng.py (the restatement the -m gpu tests
compare the HIP kernels with) against the committed known-answer vectors -- RFC 8439 section 2.3.2 and keystreams from OpenSSL's
independent implementation (tests/golden/gen_chacha_kat.py) -- and the documented use of the keystream per kind of draw."""
import json
import os

from conftest import GOLDEN
from oracle import chacha_rng as c


def test_block_function_known_answers():
    vectors = json.load(open(os.path.join(GOLDEN, "chacha20_kat.json")))
    assert len(vectors) >= 6 and vectors[0]["source"].startswith("RFC 8439")
    for v in vectors:
        key, nonce, ks = bytes.fromhex(v["key"]), bytes.fromhex(v["nonce"]), bytes.fromhex(v["keystream"])
        got = b"".join(c.chacha20_block(key, (v["counter"] + j) & 0xFFFFFFFF, nonce) for j in range(len(ks) // 64))
        assert got == ks, v["source"]


def test_rfc_state_words():
    # the same vector as 16 little-endian words (RFC 8439 2.3.2, "ChaCha state at the end of the ChaCha20 operation")
    w = c.chacha20_block_words(bytes(range(32)), 1, (0x09000000, 0x4A000000, 0))
    assert w[:4] == [0xE4E7F110, 0x15593BD1, 0x1FDD0F50, 0xC47120A3] and w[12:] == [0xD19C12B5, 0xB94E16DE, 0xE883D0CB, 0x4E3C50A2]


def test_draw_kinds_use_the_keystream_as_documented():
    key = bytes(range(32))
    def words(item, call, n):
        ws = c.WordStream(key, item, call)
        return [ws.next() for _ in range(n)]

    # bits: first words of the item's stream, top word masked
    assert c.rng_bits(key, 3, 40, 2) == [(w[0] | (w[1] << 32)) & ((1 << 40) - 1) for w in (words(0, 3, 2), words(1, 3, 2))]
    # the item's stream starts at block counter 0 with nonce (item, call_lo, call_hi) and continues into block 1
    assert words(7, (5 << 32) | 9, 20) == c.chacha20_block_words(key, 0, (7, 9, 5)) + c.chacha20_block_words(key, 1, (7, 9, 5))[:4]
    # below: attempts are consecutive word groups; the first candidate under n wins
    n = (1 << 63) + 12345
    for i, v in enumerate(c.rng_below(key, 1, n, 50, nonzero=True)):
        w = words(i, 1, 64)
        cands = [(w[2 * t] | (w[2 * t + 1] << 32)) for t in range(32)]
        assert v == next(x for x in cands if 0 < x < n)
    small = c.rng_below(key, 2, 5, 400)
    assert set(small) == {0, 1, 2, 3, 4} and set(c.rng_below(key, 2, 5, 400, nonzero=True)) == {1, 2, 3, 4}
    # coins: the keystream of item i // 512 read bit by bit
    coins = c.rng_coins(key, 4, 1200)
    blk = [c.chacha20_block_words(key, 0, (b, 4, 0)) for b in range(3)]
    assert coins == [(blk[i // 512][(i // 32) % 16] >> (i % 32)) & 1 for i in range(1200)]
    # permutations: valid, item-wise independent of the batch size, and all positions move
    perms = c.rng_permutations(key, 6, 33, 40)
    assert all(sorted(p) == list(range(33)) for p in perms) and perms[:7] == c.rng_permutations(key, 6, 33, 7)
    assert len({tuple(p) for p in perms}) == 40 and c.rng_permutations(key, 6, 1, 2) == [[0], [0]]
    # a call number of its own for every call: equal parameters, different streams
    assert c.rng_bits(key, 0, 64, 4) != c.rng_bits(key, 1, 64, 4)
