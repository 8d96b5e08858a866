"""Dev tool: A/B two builds of the library on the same box: runs bench.py alternately with SC_AMD_LIB set to each."""
import json, os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rep in range(3):
    for l in libs:
        env = dict(os.environ, SC_AMD_LIB=os.path.abspath(l))
        cp = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-extras", "--steps", "5", "--warmup", "2"] + os.environ.get("AB_ARGS", "").split(), env=env, capture_output=True, text=True)
        d = json.loads(cp.stdout.strip().splitlines()[-1])
        res[l].append(d["value"])
        print(os.path.basename(l), rep, f"{d['value']:.0f}", f"{d['roofline']['launch_ms']:.2f} ms", flush=True)
for l in libs:
    print(os.path.basename(l), "best", f"{max(res[l]):.0f}", "mean", f"{sum(res[l])/len(res[l]):.0f}")
