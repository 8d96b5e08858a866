"""Dev tool: where one batch step's device time goes, from a rocprofv3 kernel-trace csv of a SINGLE-STREAM run
(`rocprofv3 --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 --streams 1 --no-extras --no-cpu-baseline`).

One step = from the second-to-last `k_plain_alice` dispatch (the first kernel of sc_initiator_step1) to the last one: a whole
period of the timed loop (what follows the last step in bench.py -- checks, the roofline launches -- stays out).  Prints device time by kernel family (library kernels `sc::*` and the step entries' small word kernels against
everything torch or the runtime launched) and the idle time between dispatches."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
LIB = ("sc::",)
is_lib = lambda n: any(t in n for t in LIB)                                             # noqa: E731
starts = [i for i, r in enumerate(rows) if "k_plain_alice" in r["Kernel_Name"]]
if len(starts) < 2:
    sys.exit("need at least two steps in the trace (--steps 2)")
step = rows[starts[-2]:starts[-1]]          # one whole period of the timed loop: step start to the next step's start
t0, t1 = int(step[0]["Start_Timestamp"]), int(rows[starts[-1]]["Start_Timestamp"])
span = (t1 - t0) / 1e6
by, cnt = defaultdict(float), defaultdict(int)
busy_end, idle = t0, 0.0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "")
    key = "library kernels (sc::* and the step entries' word kernels)" if is_lib(name) else name[:name.find("(")][:80] if "(" in name else name[:80]
    by[key] += (e - s) / 1e6
    cnt[key] += 1
    if s > busy_end:
        idle += (s - busy_end) / 1e6
    busy_end = max(busy_end, e)
print(f"last step: {len(step)} dispatches spanning {span:.1f} ms")
other = 0.0
for k, v in sorted(by.items(), key=lambda kv: -kv[1]):
    print(f"{k:86s} {v:9.3f} ms  {100 * v / span:7.3f} %  ({cnt[k]} dispatches)")
    if not k.startswith("library"):
        other += v
print(f"{'all non-library kernels':86s} {other:9.3f} ms  {100 * other / span:7.3f} %")
print(f"{'some kernel running (union of the dispatches; a context that owns the chip forks its CRT halves)':86s} {span - idle:9.3f} ms  {100 * (span - idle) / span:7.3f} %")
print(f"{'idle between dispatches':86s} {idle:9.3f} ms  {100 * idle / span:7.3f} %")
