#!/bin/bash
# kernel timeline of one configs[1] step (two library contexts) for two builds: where the critical path waits
set -o pipefail
OUT=gpurun_out/${1:-r04t}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ARGS="bench.py --steps 4 --warmup 2 --batch 4096 --l 16 --dgk dgk_2048_l16 --no-extras --no-cpu-baseline --no-other-configs"
for lib in protocols/secure_comparison_amd/libsc_amd.so build_ab/libsc_step4_tables.so; do
  tag=$(basename $lib .so)
  export SC_AMD_LIB=$PWD/$lib
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/$tag -o t -- python3 $ARGS > $OUT/$tag.log 2>&1 || { tail -5 $OUT/$tag.log; exit 1; }
  python3 - $(find $OUT/$tag -name "*kernel_trace.csv" | head -1) > $OUT/$tag.timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_plain_alice' in r['Kernel_Name']]
i0, i1 = idx[-2], idx[-1]
t0 = int(rows[i0]['Start_Timestamp'])
def short(n):
    n = n.replace('void ', '').replace('sc::', '')
    return n[:n.find('(')][:44] if '(' in n else n[:44]
print("step span ms", (int(rows[i1]['Start_Timestamp']) - t0) / 1e6)
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if (e - s) > 150e3: print(f"q{r['Queue_Id']:>2} {(s-t0)/1e6:8.2f} .. {(e-t0)/1e6:8.2f}  dur {(e-s)/1e6:7.2f} ms grid {r['Grid_Size_X']:>7} {short(r['Kernel_Name'])}")
PY
  tail -1 $OUT/$tag.log | cut -c1-200
done
