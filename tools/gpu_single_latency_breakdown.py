"""Dev tool: where the wall time of ONE interactive comparison (BASELINE configs[0]) goes: wall-clock per wrapped call, summed
over both parties, median of a few runs."""
import asyncio, json, os, sys, time, statistics, collections
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import bench
from protocols.secure_comparison_amd import DGK, Initiator, KeyHolder, Paillier
from protocols.secure_comparison_amd.communicator import InMemoryCommunicator
from protocols.secure_comparison_amd.schemes import default_engine
from protocols.secure_comparison_amd import schemes as S

keys = json.load(open(bench.KEYS))
pj, dj = keys["paillier_1024"], keys["dgk_1024_l16"]
p, q, l = int(pj["p"], 16), int(pj["q"], 16), 16
H = lambda name: int(dj[name], 16)  # noqa: E731
eng = default_engine()
bob_p = Paillier(p * q, p, q, engine=eng)
bob_d = DGK(H("p") * H("q"), H("g"), H("h"), H("u"), dj["t"], H("p"), H("q"), H("v_p"), H("v_q"), engine=eng, randomizer_bits=400)
acc = collections.defaultdict(float)


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            torch.cuda.synchronize()
            acc[label] += (time.perf_counter() - t0) * 1e3
    setattr(obj, name, w)


wrap(S.Paillier, "boot_randomness_generation", "boot paillier randomizers (1 + 3)")
wrap(S.DGK, "boot_randomness_generation", "boot dgk randomizers (17 + 17)")
for cls, names in ((Initiator, ("step_1_batch", "step_4_batch", "step_6_7_batch")), (KeyHolder, ("step_2_4b_batch", "step_4j_5_batch"))):
    for n in names:
        f = getattr(cls, n)
        def mk(f=f, n=n):
            def w(*a, **k):
                t0 = time.perf_counter()
                try:
                    return f(*a, **k)
                finally:
                    torch.cuda.synchronize()
                    acc[n] += (time.perf_counter() - t0) * 1e3
            return staticmethod(w)
        setattr(cls, n, mk())
wrap(type(eng), "upload", "engine.upload")
wrap(type(eng), "download", "engine.download")
wrap(S.Paillier, "unsafe_encrypt", "unsafe_encrypt x, y")


def once():
    comm = InMemoryCommunicator()
    alice, bob = Initiator(l, comm, "keyholder"), KeyHolder(l, comm.peer(), "initiator", bob_p, bob_d)

    async def go():
        res, _ = await asyncio.gather(alice.perform_secure_comparison(23, 42), bob.perform_secure_comparison())
        return res
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = asyncio.run(go())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


once(); once()
acc.clear()
N = 7
tot = [once() for _ in range(N)]
print(f"total per comparison (every wrapped call followed by a device synchronisation): median {statistics.median(tot):.2f} ms")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:40s} {v / N:7.2f} ms")
print(f"  {'(sum of the above; uploads / downloads also counted inside nothing else)':40s} {sum(acc.values()) / N:7.2f} ms")
