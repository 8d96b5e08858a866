#!/bin/bash
# full GPU suite and a soak of the headline configuration on the build with the conversion-free inversion and step 4
set -o pipefail
OUT=gpurun_out/${1:-r04q}
mkdir -p $OUT
echo "== full GPU suite"; timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/tests_full.txt 2>&1; rc=$?; tail -3 $OUT/tests_full.txt; [ $rc -eq 0 ] || exit $rc
echo "== soak, headline configuration"; timeout -k 10 260 python3 tests/soak_gpu.py 150 41 > $OUT/soak.txt 2>&1; rc=$?; tail -3 $OUT/soak.txt; [ $rc -eq 0 ] || exit $rc
echo "== soak, configs[4] share"; timeout -k 10 200 python3 tests/soak_gpu.py 90 43 32768 3072 dgk_3072_l64 64 > $OUT/soak_cfg4.txt 2>&1; rc=$?; tail -3 $OUT/soak_cfg4.txt; exit $rc
