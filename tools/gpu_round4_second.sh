#!/bin/bash
# Round 4, second GPU call: the new tests, the full bench line, the two-shard and single-stream kernel traces of a step.
set -o pipefail
OUT=gpurun_out/${1:-r04b}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "== round-4 gpu tests"; timeout -k 10 500 python3 -m pytest tests/test_gpu_round4.py -x -q > $OUT/tests_r4.txt 2>&1; tail -15 $OUT/tests_r4.txt
echo "== bench (default)"; timeout -k 10 500 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
python3 tools/bench_brief.py < $OUT/bench_default.json
BENCH="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-other-configs"
echo "== kernel trace, two shards"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -o t -- python3 $BENCH > $OUT/trace2.log 2>&1 || { tail -5 $OUT/trace2.log; exit 1; }
python3 tools/step_dispatches.py $(find $OUT/trace2 -name "*kernel_trace.csv" | head -1) 2 > $OUT/step_dispatches_two_shards.txt; tail -25 $OUT/step_dispatches_two_shards.txt
echo "== kernel trace, one stream"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -o t -- python3 $BENCH --streams 1 > $OUT/trace1.log 2>&1 || { tail -5 $OUT/trace1.log; exit 1; }
python3 tools/step_dispatches.py $(find $OUT/trace1 -name "*kernel_trace.csv" | head -1) 1 > $OUT/step_dispatches_one_stream.txt; head -30 $OUT/step_dispatches_one_stream.txt
python3 tools/step_glue.py $(find $OUT/trace1 -name "*kernel_trace.csv" | head -1) > $OUT/step_glue_single_stream.txt; cat $OUT/step_glue_single_stream.txt
echo "== kernel rates"; timeout -k 10 300 python3 tools/gpu_kernel_rates.py > $OUT/kernel_rates.txt 2>&1; cat $OUT/kernel_rates.txt
