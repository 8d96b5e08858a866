"""Dev tool: A/B whole source trees on one box -- each tree's own bench.py, library and defaults -- alternating, three rounds.

    python tools/gpu_ab_trees.py NAME=DIR[:extra bench args] ...      e.g.  head=. r02=build_ab/r02 head_w20=.:--fb-window=20

Every run is `python bench.py --steps S --warmup W --no-cpu-baseline --no-extras` (+ the extra arguments) in the tree's directory;
prints value, ms per step and the dominant launch per run, then best / mean per tree.  Trees are exported with `git archive <tag> |
tar -x -C build_ab/<name>` and built there (`python -m protocols.secure_comparison_amd.build`) before the gpurun call."""
import json
import os
import subprocess
import sys

specs = []
for a in sys.argv[1:]:
    name, rest = a.split("=", 1)
    d, _, extra = rest.partition(":")
    specs.append((name, os.path.abspath(d), extra.split() if extra else []))
steps, warmup, rounds = os.environ.get("AB_STEPS", "10"), os.environ.get("AB_WARMUP", "3"), int(os.environ.get("AB_ROUNDS", "3"))
res = {n: [] for n, _, _ in specs}
for rep in range(rounds):
    for name, d, extra in specs:
        cmd = [sys.executable, "bench.py", "--steps", steps, "--warmup", warmup, "--no-cpu-baseline", "--no-extras"] + extra
        env = {k: v for k, v in os.environ.items() if k != "SC_AMD_LIB"}
        cp = subprocess.run(cmd, cwd=d, env=env, capture_output=True, text=True)
        if cp.returncode != 0:
            print(name, rep, "FAILED", cp.stderr[-800:], flush=True)
            continue
        line = json.loads(cp.stdout.strip().splitlines()[-1])
        res[name].append(line["value"])
        rl = line.get("roofline", {})
        print(f"{name:12s} round {rep}: {line['value']:9.0f} /s  {line['ms_per_step']:7.2f} ms/step  dominant launch {rl.get('launch_ms', 0):7.2f} ms"
              f"  probe {rl.get('probe_peak', 0):6.2f} T", flush=True)
for name, vals in res.items():
    if vals:
        print(f"{name:12s} best {max(vals):9.0f}  mean {sum(vals) / len(vals):9.0f}  ({len(vals)} runs)")
