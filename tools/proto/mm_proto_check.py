import struct, sys
G, L = int(sys.argv[1]), int(sys.argv[2]); W = 29; S = G * L
lines = open("mm_expect.txt").read().split()
n = int(lines[0]); exp = [int(x) for x in lines[1:]]
data = open(sys.argv[3], "rb").read()
bad = 0
for i, e in enumerate(exp):
    lim = struct.unpack_from(f"<{S}I", data, i * S * 4)
    v = sum(l << (W * k) for k, l in enumerate(lim))
    if v % n != e or v >= 2 * n: bad += 1
print("checked", len(exp), "bad", bad)
