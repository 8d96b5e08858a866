cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
OUT=gpurun_out/r04j; mkdir -p $OUT
for d in d2h h2d; do for off in 0 36 64 256 4096; do
  rm -rf $OUT/cp; timeout -k 10 120 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/cp -o p -- python3 tools/gpu_copy_probe.py $d $off > $OUT/cp.log 2>&1
  k=$(grep -c copyBuffer $OUT/cp/*kernel_trace.csv 2>/dev/null); m=$(grep -c MEMORY_COPY $OUT/cp/*memory_copy_trace.csv 2>/dev/null)
  echo "$d offset $off: copyBuffer kernels $k, SDMA copy records $m"
done; done
