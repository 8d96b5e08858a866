#!/bin/bash
# kernel timeline of single interactive comparisons (configs[0]): which launches, how long, how much of the wall time is GPU
set -o pipefail
OUT=gpurun_out/${1:-single_trace}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python3 tools/gpu_latency_single.py > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - $(find $OUT/t -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last interactive comparison of the first leg run: find the last k_plain_alice before the operator-path runs is hard; take the
# window of 9 ms before the 20-th k_plain_alice-free gap... simpler: print the dispatches between two consecutive k_plain_alice
idx = [i for i, r in enumerate(rows) if 'k_plain_alice' in r['Kernel_Name']]
i0, i1 = idx[6], idx[7]
t0 = int(rows[i0]['Start_Timestamp'])
def short(n):
    n = n.replace('void ', '').replace('sc::', '')
    return n[:n.find('(')][:44] if '(' in n else n[:44]
busy = 0
print("from one comparison's step 1 to the next one's: span ms", (int(rows[i1]['Start_Timestamp']) - t0) / 1e6, "dispatches", i1 - i0)
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy += e - s
    if (e - s) > 30e3: print(f"q{r['Queue_Id']:>2} {(s-t0)/1e6:8.3f} dur {(e-s)/1e3:8.1f} us grid {r['Grid_Size_X']:>6} {short(r['Kernel_Name'])}")
print("sum of kernel durations ms", busy / 1e6)
PY
