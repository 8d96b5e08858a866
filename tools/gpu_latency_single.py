"""Dev tool: bench.py's latency_single leg alone (BASELINE configs[0]), with the key holder's background randomizer generation on / off."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import bench
from protocols.secure_comparison_amd import KeyHolder
from protocols.secure_comparison_amd.schemes import default_engine

keys = json.load(open(bench.KEYS))
for bg in (True, False, True):
    KeyHolder.background_randomness = bg
    r = bench.latency_single_leg(torch, default_engine(), keys)
    print("background", bg, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if k.endswith("_ms") or k == "correct"}, flush=True)
