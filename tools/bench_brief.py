"""Dev tool: print the headline fields of bench.py's JSON line read from stdin (label as argv[1])."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
rl = d["roofline"]
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["value"]), "cmp/s", round(d["ms_per_step"], 1), "ms", "step frac", round(d["roofline_whole_step"]["frac"], 3),
      "dominant frac", round(rl["frac"], 3), "streams", d["config"]["streams_per_gpu"], flush=True)
if "step_ms" in d:
    print("  step ms min/median/max", *(round(d["step_ms"][k], 1) for k in ("min", "median", "max")))
if "launch_ms_before" in rl:
    print("  before/after: launch", round(rl["launch_ms_before"], 2), round(rl["launch_ms_after"], 2), "ms; probe", round(rl["probe_peak_before"], 2), round(rl["probe_peak_after"], 2),
          "T; clock", rl["clock_ghz_before"], rl["clock_ghz_after"], "GHz")
for k in ("policy", "latency_single"):
    if k in d:
        print(" ", k, json.dumps(d[k])[:400])
for o in d.get("other_configs", []):
    print("  other:", json.dumps(o)[:300])
ip = d.get("interactive_protocol")
if ip:
    print("  interactive", round(ip["value"]), "ratio", round(ip["ratio_to_headline"], 3), "byte transport", round(ip["byte_transport"]["value"]), {k: (round(v["value"]), round(v.get("whole_run_value", 0))) for k, v in ip["byte_transport"].get("pipelined", {}).items() if "value" in v},
          "unpipelined", round(ip["byte_transport"].get("single_session_unpipelined", {}).get("value", 0)))
if "cpu_baseline" in d:
    print("  cpu", d["cpu_baseline"].get("value"), d["cpu_baseline"].get("sample", "")[:120])
