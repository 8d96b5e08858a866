#!/bin/bash
# final checks of the round: default bench run on a fresh box, full GPU suite, smoke, quickstart, a short soak
set -o pipefail
OUT=gpurun_out/${1:-r04final2}
mkdir -p $OUT
echo "== default bench run"; timeout -k 10 700 python3 bench.py > $OUT/bench_default_run.json 2> $OUT/bench_default_run.err; rc=$?; tail -c 300 $OUT/bench_default_run.json; echo; [ $rc -eq 0 ] || { tail -5 $OUT/bench_default_run.err; exit $rc; }
echo "== full GPU suite"; timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/tests_full.txt 2>&1; rc=$?; tail -3 $OUT/tests_full.txt; [ $rc -eq 0 ] || exit $rc
echo "== smoke"; timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== quickstart"; timeout -k 10 200 python3 examples/quickstart.py 2>&1 | tail -3
echo "== soak"; timeout -k 10 200 python3 tests/soak_gpu.py 90 47 > $OUT/soak.txt 2>&1; tail -1 $OUT/soak.txt
