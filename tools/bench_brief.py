"""Dev tool: print the headline fields of bench.py's JSON line read from stdin (label as argv[1])."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["value"]), "cmp/s", round(d["ms_per_step"], 1), "ms", "step frac", round(d["roofline_whole_step"]["frac"], 3),
      "dominant frac", round(d["roofline"]["frac"], 3), "streams", d["config"]["streams_per_gpu"], flush=True)
