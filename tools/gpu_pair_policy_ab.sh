#!/bin/bash
# A/B of the pair-launch segment policy (sc_ctx_set_pair_policy, environment SC_PAIR_HOLD_MS at context creation) on one box, alternated:
# hold 0 = never cut (whole launches), 13 ms ~ round 4's shape-keyed choice for the (4,18) launch (four segments, the key holder's
# 14-ms rounds whole), 5 ms = the default.  Headline (B = 65536), the configs[3] share (B = 131072) and the configs[4] share.
# usage: tools/gpu_pair_policy_ab.sh OUT.txt
out=${1:-gpurun_out/pair_policy_ab.txt}
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = d["step_ms"]
    print("%8.0f /s   step ms %.1f / %.1f / %.1f   whole-step %.4f" % (d["value"], s["min"], s["median"], s["max"], d["roofline_whole_step"]["frac"]))
except Exception as e:
    print("no line:", e)
PY
}
for rep in 1 2; do
  for hold in 0 13 7 5; do
    SC_PAIR_HOLD_MS=$hold python bench.py --steps 10 --warmup 2 --no-extras --no-other-configs --no-cpu-baseline > /tmp/ab.json 2> /tmp/ab.err || tail -3 /tmp/ab.err >> $out
    echo "B=65536  hold_ms=$hold  $(line /tmp/ab.json)" >> $out
  done
done
for hold in 0 5; do
  SC_PAIR_HOLD_MS=$hold python bench.py --batch 131072 --steps 4 --warmup 1 --no-extras --no-other-configs --no-cpu-baseline > /tmp/ab.json 2> /tmp/ab.err || tail -3 /tmp/ab.err >> $out
  echo "B=131072 hold_ms=$hold  $(line /tmp/ab.json)" >> $out
done
for hold in 0 13 5; do
  SC_PAIR_HOLD_MS=$hold python bench.py --batch 32768 --l 64 --pbits 3072 --dgk dgk_2048_l64 --steps 4 --warmup 1 --no-extras --no-other-configs --no-cpu-baseline > /tmp/ab.json 2> /tmp/ab.err || tail -3 /tmp/ab.err >> $out
  echo "cfg4 share hold_ms=$hold  $(line /tmp/ab.json)" >> $out
done
