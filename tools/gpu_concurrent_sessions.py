"""Throughput of many concurrent single comparisons through the session coalescer (bench.concurrent_sessions_leg), plus a cProfile
of one run: where the wall clock of N sessions goes (GPU calls, integer <-> word conversion, asyncio, ciphertext objects).
usage: python tools/gpu_concurrent_sessions.py [sessions] [--profile]"""
import cProfile
import json
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from protocols.secure_comparison_amd.schemes import default_engine  # noqa: E402

sessions = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1024
keys = json.load(open(bench.KEYS))
eng = default_engine()
out = bench.concurrent_sessions_leg(torch, eng, keys, sessions=sessions)
for row in out["shapes"]:
    print(json.dumps(row))
if "--gc" in sys.argv:       # how much of the wall clock is the cyclic collector walking the interpreter's long-lived objects
    import gc

    gc.collect()
    gc.freeze()
    for row in bench.concurrent_sessions_leg(torch, eng, keys, sessions=sessions)["shapes"]:
        print("gc.freeze():", round(row["value"]), "comparisons/s,", row["workload"])
    gc.disable()
    for row in bench.concurrent_sessions_leg(torch, eng, keys, sessions=sessions)["shapes"]:
        print("gc.disable():", round(row["value"]), "comparisons/s,", row["workload"])
    gc.enable()
    gc.unfreeze()
if "--profile" in sys.argv:
    for shape in (("paillier_1024", "dgk_1024_l16", 16), ("paillier_2048", "dgk_2048_l32", 32)):
        pr = cProfile.Profile()
        pr.enable()
        bench.concurrent_sessions_leg(torch, eng, keys, sessions=sessions, shapes=(shape,))
        pr.disable()
        print("==== profile", shape)
        pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
        pstats.Stats(pr).sort_stats("tottime").print_stats(30)
