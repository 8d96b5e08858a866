"""Generates tests/golden/chacha20_kat.json: known-answer vectors for the cipher core of the device-side generator.

 * the RFC 8439 section 2.3.2 block-function test vector (key 00..1f, counter 1, nonce 000000090000004a00000000), typed in from
   the RFC;
 * keystreams from OpenSSL's independent ChaCha20 (`openssl enc -chacha20`, whose 16-byte IV is the 32-bit little-endian block
   counter followed by the 12-byte nonce) for a few keys / counters / nonces, including nonce words of the shape the library uses
   (item, call_lo, call_hi) and a counter that is not zero.

Run in the build container (needs the `openssl` command line tool); the JSON is committed, the tests never run openssl."""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))

RFC_2_3_2 = {
    "source": "RFC 8439 section 2.3.2",
    "key": bytes(range(32)).hex(), "counter": 1, "nonce": "000000090000004a00000000",
    "keystream": "10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
                 "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e",
}


def openssl_keystream(key: bytes, counter: int, nonce: bytes, nbytes: int) -> bytes:
    iv = counter.to_bytes(4, "little") + nonce
    cp = subprocess.run(["openssl", "enc", "-chacha20", "-K", key.hex(), "-iv", iv.hex(), "-nosalt"], input=bytes(nbytes),
                        capture_output=True, check=True)
    assert len(cp.stdout) == nbytes
    return cp.stdout


def main() -> None:
    import hashlib

    cases = [RFC_2_3_2]
    for t, (counter, nonce_words, nblocks) in enumerate([(0, (0, 0, 0), 2), (0, (5, 7, 0), 4), (3, (0xFFFFFFFF, 0x12345678, 1), 2),
                                                         (0, (65535, 2, 0), 5), (0xFFFFFFFE, (1, 2, 3), 1)]):
        key = hashlib.sha256(b"sc-amd chacha kat %d" % t).digest()
        nonce = b"".join(w.to_bytes(4, "little") for w in nonce_words)
        ks = openssl_keystream(key, counter, nonce, 64 * nblocks)
        cases.append({"source": "openssl enc -chacha20 (OpenSSL 3.0.2)", "key": key.hex(), "counter": counter, "nonce": nonce.hex(),
                      "keystream": ks.hex()})
    with open(os.path.join(HERE, "chacha20_kat.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print("wrote %d vectors" % len(cases))


if __name__ == "__main__":
    main()
