"""Experiment: two shards of a SMALL batch on two HIP streams restricted to complementary halves of the CUs
(hipExtStreamCreateWithCUMask), so that two under-filled launches do not share SIMDs."""
import ctypes as C, json, os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import bench
from protocols.secure_comparison_amd import DGK, Paillier
from protocols.secure_comparison_amd.batch import secure_comparison_batch
from protocols.secure_comparison_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
MODE = sys.argv[2] if len(sys.argv) > 2 else "masked"      # masked | plain | single
l, rbits = 16, 400
torch.cuda.init(); torch.zeros(1, device="cuda")
hip = C.CDLL("libamdhip64.so")
ncu = torch.cuda.get_device_properties(0).multi_processor_count

def masked_stream(bits):
    words = (ncu + 31) // 32
    arr = (C.c_uint32 * words)()
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

NS = 1 if MODE == "single" else 2
if MODE == "masked":
    half = sys.argv[3] if len(sys.argv) > 3 else "interleave"
    if half == "interleave":
        sets = [[c for c in range(ncu) if c % 2 == k] for k in range(2)]
    else:
        sets = [list(range(0, ncu // 2)), list(range(ncu // 2, ncu))]
    streams = [masked_stream(s) for s in sets]
else:
    streams = [torch.cuda.Stream() for _ in range(NS)]
keys = json.load(open(bench.KEYS))
pj, dj = keys["paillier_2048"], keys["dgk_2048_l16"]
p, q = int(pj["p"], 16), int(pj["q"], 16)
H = lambda k: int(dj[k], 16)
parts = []
for i in range(NS):
    eng = Engine()
    bob_p = Paillier(p * q, p, q, engine=eng); alice_p = bob_p.public_copy()
    bob_d = DGK(H("p") * H("q"), H("g"), H("h"), H("u"), dj["t"], H("p"), H("q"), H("v_p"), H("v_q"), engine=eng, randomizer_bits=rbits, fixed_base_window=20)
    alice_d = bob_d.public_copy(); bob_d.prepare(), alice_d.prepare()
    x, y, xe, ye, dr = bench.synth_inputs(eng, l, alice_p, bob_p, bob_d, B // NS, rbits, i)
    parts.append(dict(ap=alice_p, ad=alice_d, bp=bob_p, bd=bob_d, x=x, y=y, xe=xe, ye=ye, dr=dr, stream=streams[i]))
torch.cuda.synchronize()

def work(pt, out, k):
    with torch.cuda.stream(pt["stream"]):
        out[k] = secure_comparison_batch(pt["xe"], pt["ye"], l, pt["ap"], pt["ad"], pt["bp"], pt["bd"], pt["dr"], randomize=True)

for rep in range(5):
    out = [None] * NS
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(pt, out, k)) for k, pt in enumerate(parts)]
    [t.start() for t in ths]; [t.join() for t in ths]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"rep {rep}: {MODE} B={B}: {dt*1e3:.1f} ms -> {B/dt:.0f} cmp/s", flush=True)
for k, pt in enumerate(parts):
    dec = pt["bp"].decrypt_raw_batch(out[k])
    assert bool((dec[:, 0] == (pt["x"] <= pt["y"]).to(torch.int32)).all().item())
print("results correct")
