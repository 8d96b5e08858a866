#!/bin/bash
# half-size fixed-base tables of the key holder at the key's window (20) instead of 16: rates and whole-step A/B
set -o pipefail
OUT=gpurun_out/${1:-r04r}
mkdir -p $OUT
LIBS="protocols/secure_comparison_amd/libsc_amd.so build_ab/libsc_r04_inv3.so"
echo "== parity (bit encryption paths)"; timeout -k 10 600 python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -x -q > $OUT/tests.txt 2>&1; rc=$?; tail -2 $OUT/tests.txt; [ $rc -eq 0 ] || exit $rc
echo "== launch-group rates"; timeout -k 10 500 python3 tools/gpu_kernel_rates.py $LIBS > $OUT/kernel_rates.txt 2>&1 && cat $OUT/kernel_rates.txt
echo "== single stream"; AB_ARGS="--no-other-configs --steps 4 --streams 1" timeout -k 10 500 python3 tools/gpu_ab.py $LIBS > $OUT/ab_single.txt 2>&1; tail -3 $OUT/ab_single.txt
echo "== headline (two shards)"; AB_ARGS="--no-other-configs --steps 8" timeout -k 10 600 python3 tools/gpu_ab.py $LIBS > $OUT/ab.txt 2>&1; tail -3 $OUT/ab.txt
