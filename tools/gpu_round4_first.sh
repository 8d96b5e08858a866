#!/bin/bash
# First GPU call of round 4: the new bench line, the A/B against the round-2 tag, the whole GPU suite.
set -o pipefail
OUT=gpurun_out/${1:-r04a}
mkdir -p $OUT
echo "== bench (default)"; timeout -k 10 400 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
python3 tools/bench_brief.py < $OUT/bench_default.json
echo "== A/B trees"; timeout -k 10 500 python3 tools/gpu_ab_trees.py head=.:--no-other-configs r02=build_ab/r02 head_w20=.:"--no-other-configs --fb-window 20" > $OUT/ab_trees.txt 2>&1; tail -12 $OUT/ab_trees.txt
echo "== gpu tests"; timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $OUT/tests.txt 2>&1; tail -15 $OUT/tests.txt
