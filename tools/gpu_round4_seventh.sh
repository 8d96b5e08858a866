#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r04k}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "== tests"; timeout -k 10 400 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -x -q -k "chunk or interactive or protocol or shard" > $OUT/tests.txt 2>&1; tail -4 $OUT/tests.txt
for cfg in "2 1 0" "2 1 200" "3 1 130" "1 1 0"; do f=$OUT/wire_$(echo $cfg | tr ' ' '_').txt; timeout -k 10 200 python3 tools/gpu_wire_probe.py $cfg 2>&1 | grep -v amdgpu.ids > $f; head -3 $f; done
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/wt -o w -- python3 tools/gpu_wire_probe.py 2 1 200 > $OUT/wt.log 2>&1; python3 tools/trace_gaps.py $OUT/wt 2.0e9 | tail -12
