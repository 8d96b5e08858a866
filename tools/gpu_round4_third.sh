#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r04c}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "== gpu tests (round 3 + 4)"; timeout -k 10 600 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -x -q > $OUT/tests.txt 2>&1; tail -5 $OUT/tests.txt
echo "== wire probe"
for cfg in "2 1 0" "2 1 180" "2 2 0" "2 2 90" "1 2 0" "4 1 90"; do timeout -k 10 200 python3 tools/gpu_wire_probe.py $cfg > $OUT/wire_$(echo $cfg | tr ' ' '_').txt 2>&1; head -1 $OUT/wire_$(echo $cfg | tr ' ' '_').txt; done
echo "== configs[1] standalone"; timeout -k 10 200 python3 bench.py --batch 4096 --l 16 --dgk dgk_2048_l16 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/bench_cfg1.json 2> $OUT/bench_cfg1.err; python3 tools/bench_brief.py < $OUT/bench_cfg1.json
BENCH="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-other-configs"
echo "== kernel trace, two shards"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -o t -- python3 $BENCH > $OUT/trace2.log 2>&1 || { tail -5 $OUT/trace2.log; exit 1; }
python3 tools/step_dispatches.py $(find $OUT/trace2 -name "*kernel_trace.csv" | head -1) 2 > $OUT/step_dispatches_two_shards.txt; tail -12 $OUT/step_dispatches_two_shards.txt
