"""Quick start on an MI355X: the two ways to use the package.

1. The reference-style single comparison: Initiator / KeyHolder coroutines over a transport
   (same flow as the usage example in the reference's README, with an in-memory transport instead of HTTP pools).
2. The batched path: thousands of comparisons per call on device arrays.

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
    await batched(paillier, dgk)


if __name__ == "__main__":
    asyncio.run(main())
