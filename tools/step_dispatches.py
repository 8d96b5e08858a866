"""Dev tool: every dispatch of ONE timed step of a two-shard (or single-stream) bench.py run, from a rocprofv3 kernel-trace csv
(`rocprofv3 --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-other-configs`).

The step window runs from the first `k_plain_alice` dispatch of the second-to-last step to the first one of the last step.  Prints
the dispatches that are not the library's interpreter / inversion kernels (name, queue, start, duration) -- the runtime's copy and
fill kernels, torch kernels, the plain word kernels -- and a summary: how many there are, the longest, and the total."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 2
starts = [i for i, r in enumerate(rows) if "k_plain_alice" in r["Kernel_Name"]]
if len(starts) < 2 * streams:
    sys.exit("need at least two steps in the trace")
i0, i1 = starts[-2 * streams], starts[-streams]
t0, t1 = int(rows[i0]["Start_Timestamp"]), int(rows[i1]["Start_Timestamp"])
BIG = ("sc::k_vm", "sc::k_pvm", "sc::k_xgcd")
small, by = [], defaultdict(lambda: [0, 0.0, 0.0])
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "")
    short = name[:name.find("(")] if "(" in name else name
    ms = (e - s) / 1e6
    fam = short if not any(b in short for b in BIG) else short.split("<")[0]
    by[fam][0] += 1
    by[fam][1] += ms
    by[fam][2] = max(by[fam][2], ms)
    if not any(b in short for b in BIG):
        small.append((ms, (s - t0) / 1e6, r["Queue_Id"], short[:70]))
print(f"one step: {i1 - i0} dispatches over {(t1 - t0) / 1e6:.1f} ms ({streams} stream(s))")
print("%-60s %6s %10s %10s" % ("kernel family", "calls", "total ms", "max ms"))
for k, (n, tot, mx) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("%-60s %6d %10.3f %10.3f" % (k[:60], n, tot, mx))
print("\ndispatches that are not interpreter / inversion kernels (duration ms, start ms, queue, name):")
for ms, at, q, name in sorted(small, key=lambda x: x[1]):
    print(f"  {ms:8.3f}  at {at:8.2f}  q{q}  {name}")
if small:
    print(f"\n{len(small)} such dispatches, longest {max(m for m, *_ in small):.3f} ms, together {sum(m for m, *_ in small):.3f} ms")
