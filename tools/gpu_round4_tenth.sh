#!/bin/bash
# which of the three changes costs the headline: launch-group rates with the step-4 and bit-encryption groups, then whole-step
# A/B on one stream and on two shards
set -o pipefail
OUT=gpurun_out/${1:-r04p}
mkdir -p $OUT
LIBS="protocols/secure_comparison_amd/libsc_amd.so build_ab/libsc_step4_tables.so build_ab/libsc_before_step4.so"
echo "== launch-group rates"; timeout -k 10 500 python3 tools/gpu_kernel_rates.py $LIBS > $OUT/kernel_rates.txt 2>&1 && cat $OUT/kernel_rates.txt
echo "== single stream"; AB_ARGS="--no-other-configs --steps 4 --streams 1" timeout -k 10 500 python3 tools/gpu_ab.py $LIBS > $OUT/ab_single.txt 2>&1; tail -12 $OUT/ab_single.txt
echo "== headline (two shards)"; AB_ARGS="--no-other-configs --steps 8" timeout -k 10 600 python3 tools/gpu_ab.py $LIBS > $OUT/ab.txt 2>&1; tail -12 $OUT/ab.txt
