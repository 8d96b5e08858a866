#!/bin/bash
# Collect the round's profiles on the GPU box (run through gpurun from the repository root):
#   kernel-trace statistics of bench.py, three separate PMC passes (FETCH_SIZE; WRITE_SIZE; the VALU / wave counters), the
#   un-profiled bench lines of every BASELINE configuration that fits one GPU.  Output under gpurun_out/$TAG/;
#   tools/summarize_profiles.py condenses it into profiles/.
set -o pipefail
TAG=${1:-r04prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
BENCH="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-other-configs"
echo "== kernel trace" && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $BENCH > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
echo "== kernel trace, single stream" && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace1 -o trace1 -- python3 $BENCH --streams 1 > $OUT/trace1.log 2>&1 || { tail -5 $OUT/trace1.log; exit 1; }
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE"; do
  name=pmcf_$(echo $set | tr ' ' '_' | cut -c1-60)
  echo "== pmc $set"
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$name -o pmc -- python3 $BENCH --streams 1 > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; exit 1; }
done
echo "== bench lines"
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_cfg2_B65536.json 2> $OUT/bench_cfg2.err || exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --batch 4096 --l 16 --dgk dgk_2048_l16 --no-extras --no-cpu-baseline > $OUT/bench_cfg1_B4096.json 2>> $OUT/bench_other.err || exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --batch 4096 --l 16 --dgk dgk_2048_l16 --no-extras --no-cpu-baseline --side-stream 0 > $OUT/bench_cfg1_B4096_one_context.json 2>> $OUT/bench_other.err || exit 1
timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --batch 131072 --no-extras --no-cpu-baseline > $OUT/bench_cfg3share_B131072.json 2>> $OUT/bench_other.err || exit 1
timeout -k 10 300 python3 bench.py --steps 4 --warmup 1 --batch 32768 --l 64 --pbits 3072 --dgk dgk_3072_l64 --no-extras --no-cpu-baseline > $OUT/bench_cfg4share_dgk3072.json 2>> $OUT/bench_other.err || exit 1
timeout -k 10 300 python3 bench.py --steps 4 --warmup 1 --batch 32768 --l 64 --pbits 3072 --dgk dgk_2048_l64 --no-extras --no-cpu-baseline > $OUT/bench_cfg4share_dgk2048.json 2>> $OUT/bench_other.err || exit 1
timeout -k 10 300 python3 tools/gpu_kernel_rates.py > $OUT/kernel_rates.txt 2>&1
echo "== two-shard step dispatches"
python3 tools/step_dispatches.py $(find $OUT/stats -name "*kernel_trace.csv" | head -1) 2 > $OUT/step_dispatches_two_shards.txt 2>&1
python3 tools/step_glue.py $(find $OUT/trace1 -name "*kernel_trace.csv" | head -1) > $OUT/step_glue_single_stream.txt 2>&1
echo done
