#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r04f}
mkdir -p $OUT
for cfg in "2 1 0" "2 1 200" "3 1 130" "2 2 0"; do f=$OUT/wire_$(echo $cfg | tr ' ' '_').txt; timeout -k 10 200 python3 tools/gpu_wire_probe.py $cfg 2>&1 | grep -v amdgpu.ids > $f; head -3 $f; done
