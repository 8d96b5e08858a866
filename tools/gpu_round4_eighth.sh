#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r04m}
mkdir -p $OUT
echo "== latency-config tests"; timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "latency or small_batch or forced or pair or fused_single" > $OUT/tests.txt 2>&1; tail -4 $OUT/tests.txt
echo "== A/B configs[1]"; AB_ARGS="--batch 4096 --l 16 --dgk dgk_2048_l16 --no-other-configs --steps 20" timeout -k 10 400 python3 tools/gpu_ab.py protocols/secure_comparison_amd/libsc_amd.so build_ab/libsc_noil.so > $OUT/ab_cfg1.txt 2>&1; tail -9 $OUT/ab_cfg1.txt
echo "== latency probe"; for lib in protocols/secure_comparison_amd/libsc_amd.so build_ab/libsc_noil.so; do SC_AMD_LIB=$PWD/$lib timeout -k 10 200 python3 - <<'PY'
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from protocols.secure_comparison_amd.schemes import default_engine
keys = json.load(open(bench.KEYS))
print(os.path.basename(os.environ["SC_AMD_LIB"]), {k: round(v, 2) if isinstance(v, float) else v for k, v in bench.latency_single_leg(torch, default_engine(), keys).items() if k.endswith("_ms")})
PY
done
