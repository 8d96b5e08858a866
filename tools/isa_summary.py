"""Instruction summary of the gfx950 kernels from the compiler's assembly listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 --cuda-device-only -S \
        protocols/secure_comparison_amd/csrc/sc_lib.hip -o /tmp/sc_lib.s
    python tools/isa_summary.py /tmp/sc_lib.s [kernel-substring ...] > profiles/rNN_isa_summary.json

Per kernel: register / scratch metadata, an instruction histogram, and the basic blocks ranked by multiply-add count (the hot
blocks are the unrolled limb-step loops) with the scratch_* (spill) instructions each block contains -- the evidence for
"the spills sit outside the hot blocks".
"""
from __future__ import annotations

import json
import re
import sys
from collections import Counter


def demangle(name: str) -> str:
    m = re.match(r"_ZN2sc(\d+)(k_\w+?)ILi(\d+)ELi(\d+)ELi(\d+)E(?:Lb([01])E)?(?:Lb([01])E)?EE", name)
    if m:
        return (f"{m.group(2)[:int(m.group(1))]}<{m.group(3)},{m.group(4)},{m.group(5)}" + (",neg1" if m.group(6) == "1" else "") +
                (",stamp" if m.group(7) == "1" else "") + ">")
    m = re.match(r"_ZN2sc(\d+)(k_\w+)", name)
    return m.group(2)[:int(m.group(1))] if m else name


def parse(path: str):
    kernels, cur, block = {}, None, None
    meta_re = re.compile(r";\s*(NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|NumSgprs|codeLenInByte|LDSByteSize)\s*:\s*(\d+)")
    for line in open(path):
        s = line.strip()
        m = re.match(r"^(_ZN2sc\w+):\s*; @", line)
        if m:
            cur = {"name": demangle(m.group(1)), "blocks": [], "meta": {}}
            kernels[cur["name"]] = cur
            block = {"label": "entry", "ins": []}
            cur["blocks"].append(block)
            continue
        if cur is None:
            continue
        mm = meta_re.search(s)
        if mm:
            cur["meta"][mm.group(1)] = int(mm.group(2))
            continue
        if s.startswith(".section") or s.startswith(".text") and cur["meta"]:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            block = {"label": m.group(1), "ins": []}
            cur["blocks"].append(block)
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        if re.match(r"^[a-z_0-9]+$", op) and block is not None and "codeLenInByte" not in cur["meta"]:
            block["ins"].append(op)
    return kernels


def summarize(k, top=6):
    hist = Counter(op for b in k["blocks"] for op in b["ins"])
    total = sum(hist.values())
    blocks = []
    for b in k["blocks"]:
        c = Counter(b["ins"])
        blocks.append({"label": b["label"], "instructions": len(b["ins"]), "v_mad_u64_u32": c.get("v_mad_u64_u32", 0),
                       "scratch": sum(v for o, v in c.items() if o.startswith("scratch_")),
                       "ds": sum(v for o, v in c.items() if o.startswith("ds_")),
                       "global": sum(v for o, v in c.items() if o.startswith("global_")),
                       "s_barrier": c.get("s_barrier", 0), "dpp": c.get("v_and_b32_dpp", 0) + c.get("v_mov_b32_dpp", 0)})
    hot = sorted(blocks, key=lambda b: -b["v_mad_u64_u32"])[:top]
    mads = hist.get("v_mad_u64_u32", 0)
    scratch_total = sum(b["scratch"] for b in blocks)
    scratch_in_hot = sum(b["scratch"] for b in blocks if b["v_mad_u64_u32"] >= 100)
    return {"kernel": k["name"], "meta": k["meta"], "instructions": total, "v_mad_u64_u32": mads,
            "scratch_instructions": scratch_total, "scratch_in_blocks_with_100+_mads": scratch_in_hot,
            "histogram_top": dict(hist.most_common(14)), "hot_blocks": hot}


def main() -> None:
    path, filters = sys.argv[1], sys.argv[2:]
    ks = parse(path)
    out = [summarize(k) for n, k in ks.items() if not filters or any(f in n for f in filters)]
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
