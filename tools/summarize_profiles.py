"""Condense rocprofv3 output directories (gpurun_out/...) into the small summaries kept under profiles/.

usage: python tools/summarize_profiles.py <round tag> <kernel-stats dir> <pmc dir> [<pmc dir> ...]

  <kernel-stats dir>  a `rocprofv3 --kernel-trace --stats --output-format csv` directory (*_kernel_stats.csv)
  <pmc dir>           one directory per `--pmc` pass (pmc_counter_collection.csv)

Writes profiles/<tag>_bench_B65536_kernel_stats.csv (kernel names shortened), profiles/<tag>_pmc_summary.csv (mean / max
per kernel, grid size and counter), profiles/<tag>_dominant_kernel_traffic.json and
profiles/<tag>_dominant_kernel_issue_analysis.json (counters of the longest dispatch of the dominant kernel).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name: str) -> str:
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    cut = name.find("(")
    if cut > 0:
        name = name[:cut]
    return name if len(name) <= 96 else name[:93] + "..."


def kernel_stats(src_dir: str, dst: str) -> str:
    src = glob.glob(os.path.join(src_dir, "*_kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                        r["MaxNs"]])
    sc = [r for r in rows if "sc::k_" in r["Name"] and "peak_probe" not in r["Name"]]
    return short(max(sc, key=lambda r: int(r["TotalDurationNs"]))["Name"])


def pmc(pmc_dirs, dominant, tag):
    summary = []
    passes = []
    merged = {}
    for d in pmc_dirs:
        name = os.path.basename(d.rstrip("/")).replace("pmcf_", "")
        per_dispatch = defaultdict(dict)
        meta = {}
        for r in csv.DictReader(open(os.path.join(d, "pmc_counter_collection.csv"))):
            did = int(r["Dispatch_Id"])
            per_dispatch[did][r["Counter_Name"]] = per_dispatch[did].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            meta[did] = (short(r["Kernel_Name"]), int(r["Grid_Size"]),
                         (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        groups = defaultdict(lambda: defaultdict(list))
        for did, ctrs in per_dispatch.items():
            k, g, _ = meta[did]
            if "sc::" not in k:
                continue
            for c, v in ctrs.items():
                groups[(k, g)][c].append(v)
        for (k, g), ctrs in sorted(groups.items()):
            for c, vals in sorted(ctrs.items()):
                summary.append([name, k, g, c, len(vals), sum(vals) / len(vals), max(vals)])
        dom = [did for did in per_dispatch if meta[did][0] == dominant]
        if dom:
            best = max(dom, key=lambda did: meta[did][2])
            entry = {"pass": name, "dur_ms": meta[best][2]}
            entry.update(per_dispatch[best])
            passes.append(entry)
            merged.update(per_dispatch[best])
            merged["dur_ms"] = meta[best][2]
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["pass", "kernel", "grid_size", "counter", "dispatches", "mean", "max"])
        w.writerows(summary)
    traffic = {
        "kernel": f"{dominant} (its longest dispatch: Alice's rho^N mod N^2 for the whole batch)",
        "source": f"rocprofv3 --pmc, separate passes, profiles/{tag}_pmc_summary.csv",
        "counters": merged,
    }
    if "FETCH_SIZE" in merged and "WRITE_SIZE" in merged:
        traffic["hbm_bytes_per_launch_uncorrected"] = (merged["FETCH_SIZE"] + merged["WRITE_SIZE"]) * 1024.0
        # gfx950: FETCH_SIZE tallies each 128-B line request at 64 B (MI355X guide, HBM section; confirmed on this library's
        # 288-B limb rows by tools/gpu_traffic_calibration.py, profiles/r01_traffic_calibration.json); WRITE_SIZE is exact
        traffic["hbm_bytes_per_launch"] = (2.0 * merged["FETCH_SIZE"] + merged["WRITE_SIZE"]) * 1024.0
    json.dump(traffic, open(os.path.join(ROOT, "profiles", f"{tag}_dominant_kernel_traffic.json"), "w"), indent=1)
    json.dump(passes, open(os.path.join(ROOT, "profiles", f"{tag}_dominant_kernel_issue_analysis.json"), "w"), indent=1)
    return merged


def main():
    tag, stats_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    dominant = kernel_stats(stats_dir, os.path.join(ROOT, "profiles", f"{tag}_bench_B65536_kernel_stats.csv"))
    print("dominant kernel:", dominant)
    merged = pmc(pmc_dirs, dominant, tag)
    for k in sorted(merged):
        print(f"  {k:28s} {merged[k]:.6g}")


if __name__ == "__main__":
    main()
