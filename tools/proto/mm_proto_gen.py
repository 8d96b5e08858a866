import random, struct, sys
G, L, count, nsq, bits = (int(a) for a in sys.argv[1:6])
W = 29; S = G * L; R = 1 << (W * S)
rng = random.Random(7)
n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
n0inv = (-pow(n, -1, 1 << W)) % (1 << W)
def limbs(x): return [(x >> (W * i)) & ((1 << W) - 1) for i in range(S)]
xs = [rng.randrange(n) for _ in range(count)]
with open("mm_in.bin", "wb") as f:
    f.write(struct.pack("<5I", G, L, count, nsq, n0inv))
    f.write(struct.pack(f"<{S}I", *limbs(n)))
    for x in xs: f.write(struct.pack(f"<{S}I", *limbs(x)))
with open("mm_expect.txt", "w") as f:
    f.write(f"{n}\n")
    Rinv = pow(R, -1, n)
    for x in xs:
        v = x
        for _ in range(nsq): v = v * v * Rinv % n
        f.write(f"{v}\n")
