"""Quick start on an MI355X: the three ways to use the package.

1. The reference-style single comparison: Initiator / KeyHolder coroutines over a transport
   (same flow as the usage example in the reference's README, with an in-memory transport instead of HTTP pools).
2. Many single comparisons at once: concurrent sessions on one player pair (the reference's test_parallel_runs shape); their steps
   are coalesced into batch launches behind the same coroutine API.
3. The batched path: thousands of comparisons per call on device arrays.

Run:  python examples/quickstart.py   (needs the GPU; builds nothing -- run `python -m protocols.secure_comparison_amd.build` first)
"""
import asyncio
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

from protocols.secure_comparison_amd import DGK, InMemoryCommunicator, Initiator, KeyHolder, Paillier  # noqa: E402
from protocols.secure_comparison_amd.keygen import next_prime  # noqa: E402
from protocols.secure_comparison_amd.randomness import uniform_below  # noqa: E402


async def single_comparison() -> None:
    l = 16
    # Bob creates the keys (small sizes so that key generation takes seconds, not a minute)
    paillier = Paillier.from_security_parameter(key_length=1024)
    dgk = DGK.from_security_parameter(v_bits=160, n_bits=1024, u=next_prime(1 << (l + 2)), full_decryption=False)
    to_bob = InMemoryCommunicator()
    alice = Initiator(l, communicator=to_bob, other_party="bob")
    bob = KeyHolder(l, communicator=to_bob.peer(), other_party="alice", scheme_paillier=paillier, scheme_dgk=dgk)
    x, y = 23, 42
    x_leq_y_enc, _ = await asyncio.gather(alice.perform_secure_comparison(paillier.unsafe_encrypt(x), paillier.unsafe_encrypt(y)),
                                          bob.perform_secure_comparison())
    print(f"[single]  {x} <= {y} :", bool(paillier.decrypt(x_leq_y_enc)))
    paillier.shut_down(), dgk.shut_down()
    return paillier, dgk


async def concurrent_singles(paillier: Paillier, dgk: DGK, sessions: int = 256) -> None:
    l = 16
    to_bob = InMemoryCommunicator()
    alice = Initiator(l, communicator=to_bob, other_party="bob")
    bob = KeyHolder(l, communicator=to_bob.peer(), other_party="alice", scheme_paillier=paillier, scheme_dgk=dgk)
    pairs = [(i * 257 % (1 << l), i * 911 % (1 << l)) for i in range(sessions)]
    t0 = time.perf_counter()
    results = await asyncio.gather(*(alice.perform_secure_comparison(x, y) for x, y in pairs),        # plaintext inputs are encrypted on the way
                                   *(bob.perform_secure_comparison() for _ in pairs))
    dt = time.perf_counter() - t0
    ok = all(bool(paillier.decrypt(r)) == (x <= y) for r, (x, y) in zip(results, pairs))
    calls = alice._coalescer().stats["calls"] + bob._coalescer().stats["calls"]
    print(f"[sessions] {sessions} concurrent single comparisons in {dt * 1e3:.0f} ms through {calls} batched library calls; all correct: {ok}")


async def batched(paillier: Paillier, dgk: DGK, count: int = 2048) -> None:
    l = 16
    dev = paillier.engine.device
    xs = torch.randint(0, 1 << l, (count,), device=dev)
    ys = torch.randint(0, 1 << l, (count,), device=dev)
    nw = paillier.mod_n.nwords

    def encrypt(v: torch.Tensor) -> torch.Tensor:
        words = torch.zeros((count, nw), dtype=torch.int32, device=dev)
        words[:, 0] = v.to(torch.int32)
        rho = uniform_below(paillier.public_key.n, count, paillier.engine, nonzero=True)     # drawn on the device
        return paillier.randomize_batch(paillier.encrypt_raw_batch(words), rho)

    to_bob = InMemoryCommunicator()
    alice = Initiator(l, communicator=to_bob, other_party="bob")
    bob = KeyHolder(l, communicator=to_bob.peer(), other_party="alice", scheme_paillier=paillier, scheme_dgk=dgk)
    x_enc, y_enc = encrypt(xs), encrypt(ys)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    result, _ = await asyncio.gather(alice.perform_secure_comparison_batch(x_enc, y_enc), bob.perform_secure_comparison_batch())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bits = paillier.decrypt_raw_batch(result)[:, 0]
    ok = bool((bits == (xs <= ys).to(torch.int32)).all().item())
    print(f"[batched] {count} comparisons in {dt * 1e3:.0f} ms incl. randomness generation and wire packing; all correct: {ok}")


async def main() -> None:
    paillier, dgk = await single_comparison()
    await concurrent_singles(paillier, dgk)
    await batched(paillier, dgk)


if __name__ == "__main__":
    asyncio.run(main())
