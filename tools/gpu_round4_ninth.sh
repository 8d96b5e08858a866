#!/bin/bash
# steps 4c-4h with per-comparison factor tables: parity subset, launch-group rates and the headline against the build before
set -o pipefail
OUT=gpurun_out/${1:-r04n}
mkdir -p $OUT
echo "== parity subset"; timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_suite.py tests/test_gpu_round3.py -m gpu -x -q > $OUT/tests.txt 2>&1; rc=$?; tail -4 $OUT/tests.txt; [ $rc -eq 0 ] || exit $rc
echo "== launch-group rates"; timeout -k 10 500 python3 tools/gpu_kernel_rates.py protocols/secure_comparison_amd/libsc_amd.so build_ab/libsc_step4_tables.so > $OUT/kernel_rates.txt 2>&1 && cat $OUT/kernel_rates.txt
echo "== headline A/B"; AB_ARGS="--no-other-configs --steps 8" timeout -k 10 500 python3 tools/gpu_ab.py protocols/secure_comparison_amd/libsc_amd.so build_ab/libsc_step4_tables.so > $OUT/ab.txt 2>&1; tail -9 $OUT/ab.txt
