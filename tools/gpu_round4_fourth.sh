#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r04d}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "== wire probe"
for cfg in "2 1 0" "2 2 0" "1 2 0" "1 1 0"; do timeout -k 10 200 python3 tools/gpu_wire_probe.py $cfg 2>&1 | grep -v amdgpu.ids > $OUT/wire_$(echo $cfg | tr ' ' '_').txt; head -1 $OUT/wire_$(echo $cfg | tr ' ' '_').txt; done
tail -3 $OUT/wire_2_1_0.txt
echo "== bench (default)"; timeout -k 10 500 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
python3 tools/bench_brief.py < $OUT/bench_default.json
echo "== kernel rates"; timeout -k 10 300 python3 tools/gpu_kernel_rates.py > $OUT/kernel_rates.txt 2>&1; cat $OUT/kernel_rates.txt
echo "== gpu tests"; timeout -k 10 800 python3 -m pytest tests -m gpu -x -q > $OUT/tests.txt 2>&1; tail -5 $OUT/tests.txt
