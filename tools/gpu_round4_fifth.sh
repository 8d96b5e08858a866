#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r04e}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/wiretrace -o w -- python3 tools/gpu_wire_probe.py 2 1 0 > $OUT/wiretrace.log 2>&1 || { tail -5 $OUT/wiretrace.log; exit 1; }
grep -v amdgpu $OUT/wiretrace.log | head -3
ls $OUT/wiretrace
python3 - <<'PY'
import csv, glob, os
out = os.environ.get("OUT", "")
f = glob.glob("gpurun_out/*/wiretrace/*memory_copy_trace.csv")[-1]
rows = list(csv.DictReader(open(f)))
print(len(rows), "copies; columns", list(rows[0].keys()))
big = [r for r in rows if int(r.get("Bytes", r.get("bytes", 0)) or 0) > 1 << 20]
print(len(big), "copies above 1 MiB")
for r in big[-24:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    b = int(r.get("Bytes", 0))
    print(r.get("Direction", ""), b >> 20, "MiB", f"{(e - s) / 1e6:8.2f} ms", f"{b / (e - s):6.1f} GB/s", "start", s)
PY
