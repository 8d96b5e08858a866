#!/bin/bash
# per-step view of ONE traced two-shard run of many steps: span, and when each shard's long launches start
set -o pipefail
OUT=gpurun_out/${1:-modes2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python3 bench.py --steps ${STEPS:-14} --warmup 1 --no-cpu-baseline --no-extras --no-other-configs > $OUT/log.txt 2>&1 || { tail -3 $OUT/log.txt; exit 1; }
python3 - $(find $OUT/t -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_plain_alice' in r['Kernel_Name']]
starts = idx[0::2]
for a, b in zip(starts[2:], starts[3:]):
    t0, t1 = int(rows[a]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])
    sig = []
    fam = collections.defaultdict(float)
    for r in rows[a:b]:
        n = r['Kernel_Name'].replace('void ', '').replace('sc::', ''); n = n[:n.find('(')]
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        fam[n] += d
        if d > 20 and 'k_pvm<4' not in n: sig.append("%s%s@%.0f+%.0f" % ({'k_vm<4, 18, 29, true>': 'BL', 'k_vm<1, 37, 28, false>': 'Z', 'k_pvm<2, 18, 29, false, false>': 'P2', 'k_vm<8, 18, 29, false>': 'V8'}.get(n, n[:8]), r['Queue_Id'], (int(r['Start_Timestamp']) - t0) / 1e6, d))
    print("span %.1f  vm8 %.0f  pvm4 %.0f  pvm2 %.0f  | %s" % ((t1 - t0) / 1e6, fam['k_vm<8, 18, 29, false>'], fam['k_pvm<4, 18, 29, true, false>'], fam['k_pvm<2, 18, 29, false, false>'], " ".join(sig)))
PY
