#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (scripts/profile_round.sh) into tracked files under profiles/:
  profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary, verbatim
  profiles/<tag>_summary.json         per-kernel calls / avg duration / PMC bytes per launch
  profiles/pmc_traffic.json           what bench.py reports as roofline.traffic
PMC handling follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected in
separate passes, are in KiB, and on gfx950 FETCH_SIZE counts half of the bytes of wide coalesced
reads -> doubled here (calibration: bsk_to_fourier reads exactly 48,627,712 B and reports 23,891 KiB).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n


def newest(pattern):   # gpurun merges into an existing directory: older passes may still lie there
    return max(glob.glob(pattern), key=os.path.getmtime)


stats_csv = newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
shutil.copy(stats_csv, os.path.join(dst, f"{tag}_kernel_stats.csv"))
kern = {}
for r in csv.DictReader(open(stats_csv)):
    kern[short(r["Name"])] = {"calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
                              "avg_ms": float(r["AverageNs"]) / 1e6, "pct": float(r["Percentage"])}


def pmc(sub, counter):
    f = newest(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
for k in kern:
    if k in fetch:
        n, v = fetch[k]
        kern[k]["fetch_bytes_per_launch"] = v / n * 1024 * 2      # gfx950 correction (x2)
        kern[k]["fetch_size_raw_kib_per_launch"] = v / n
    if k in write:
        n, v = write[k]
        kern[k]["write_bytes_per_launch"] = v / n * 1024
bench_line = None
p = os.path.join(src, "bench_line.json")
if os.path.exists(p) and os.path.getsize(p):
    bench_line = json.loads(open(p).read().strip().splitlines()[-1])
# cross-check with bench.py's own HIP-event timing: the profiled command runs 1 warm-up + `steps`
# timed batched steps + 1 PCIe-inclusive step (plus a few single-find launches), bench.py reports
# launches * launch_ms for the `steps` timed ones
check = None
if bench_line and "roofline" in bench_line:
    rl = bench_line["roofline"]
    steps = bench_line.get("steps", 1)
    warm = bench_line.get("warmup", 0)
    extra_passes = 1 if "pcie_inclusive" in bench_line else 0
    bench_total = rl["launches"] * rl["launch_ms"] * (steps + warm + extra_passes) / max(steps, 1)
    dom = max((k for k in kern if "blind_rotate" in k), key=lambda k: kern[k]["total_ms"])
    prof_total = kern[dom]["total_ms"]
    check = {"bench_hip_event_total_ms_scaled_to_all_batched_runs": round(bench_total, 1),
             "rocprof_dominant_blind_rotate_kernel": dom,
             "rocprof_total_ms": round(prof_total, 1),
             "ratio": round(prof_total / bench_total, 4) if bench_total else None}
extra = sys.argv[2] if len(sys.argv) > 2 else ""      # e.g. "--params multibit2"
out = {"tag": tag, "command": f"python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline {extra}".strip() + " (under rocprofv3)",
       "duration_cross_check": check,
       "kernels": kern, "bench_line_under_profiler": bench_line}
json.dump(out, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
br = sorted((k for k in kern if "blind_rotate" in k), key=lambda k: -kern[k]["total_ms"])
if br:
    k = kern[br[0]]
    traffic = {"source": f"profiles/{tag}_summary.json", "kernel": br[0],
               "blind_rotate_hbm_bytes_per_launch": round(k.get("fetch_bytes_per_launch", 0) + k.get("write_bytes_per_launch", 0)),
               "note": "FETCH_SIZE x2 (gfx950) + WRITE_SIZE, KiB -> bytes, averaged over the launches of one bench run"}
    params = extra.split()[-1] if "--params" in extra else "classic"
    name = "pmc_traffic.json" if params == "classic" else f"pmc_traffic_{params}.json"
    json.dump(traffic, open(os.path.join(dst, name), "w"), indent=1)
for k, v in kern.items():
    print(f"{k[:70]:70s} calls={v['calls']:5d} avg={v['avg_ms']:9.4f} ms  fetch/launch={v.get('fetch_bytes_per_launch', 0)/1e6:10.2f} MB")
