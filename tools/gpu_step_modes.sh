#!/bin/bash
# the two-shard step settles at ~354 or ~362 ms per run: per-kernel-family device time of one step in several traced runs
set -o pipefail
OUT=gpurun_out/${1:-modes}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for rep in ${REPS:-1 2 3 4 5}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/t$rep -o t -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-other-configs > $OUT/log$rep.txt 2>&1 || { tail -3 $OUT/log$rep.txt; exit 1; }
  python3 - $(find $OUT/t$rep -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_plain_alice' in r['Kernel_Name']]
i0, i1 = idx[-4], idx[-2]
t0, t1 = int(rows[i0]['Start_Timestamp']), int(rows[i1]['Start_Timestamp'])
fam = collections.defaultdict(float); big = []
for r in rows[i0:i1]:
    n = r['Kernel_Name'].replace('void ', '').replace('sc::', ''); n = n[:n.find('(')]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    fam[n] += d
    if d > 20: big.append((round((int(r['Start_Timestamp']) - t0) / 1e6, 1), round(d, 1), n[:24], r['Queue_Id']))
print("step span %.1f ms |" % ((t1 - t0) / 1e6), " ".join("%s %.1f" % (k.replace(', 29, false', '').replace(', false', ''), v) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:7]))
print("   launches > 20 ms (start, dur, kernel, queue):", big)
PY
done
