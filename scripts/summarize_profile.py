#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (scripts/profile_round.sh) into tracked files under profiles/:
  profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary, verbatim
  profiles/<tag>_summary.json         per-kernel calls / avg duration / PMC bytes per launch
  profiles/pmc_traffic.json           what bench.py reports as roofline.traffic
  profiles/<tag>_issue_rates.jsonl    scripts/micro/f64_rate: cycles per wave-instruction, clock under load
  profiles/compute_roofline.json      SQ counters of the dominant blind-rotate kernel per wave and CMUX
                                      iteration + issue costs: what bench.py turns into roofline_compute
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


def boots_of(name):
    """bootstraps per workgroup: 2 for blind_rotate_wpp_kernel<BL, STAMP, ABL, 2> (two bootstraps share a workgroup), else 1"""
    import re
    m = re.search(r"blind_rotate_wpp_kernel<[^>]*,\s*(\d+)>", name)
    return int(m.group(1)) if m else 1


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


BATCHED_WGS = 512     # bench.py's timed launches are >= 512 workgroups; the untimed single-find launches are 2-7


def pmc(sub, counter, min_wgs=0):
    """(launches, sum) per kernel; min_wgs > 0 keeps the dispatches of the batched steps only -- the basis of
    bench.py's `launch_ms` (HIP events around the batched launches)"""
    f = newest(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        if min_wgs and float(r.get("Grid_Size", 0)) / max(float(r.get("Workgroup_Size", 1)), 1) < min_wgs:
            continue
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
fetch_b, write_b = pmc("fetch", "FETCH_SIZE", BATCHED_WGS), pmc("write", "WRITE_SIZE", BATCHED_WGS)
for k in kern:
    if k in fetch:
        n, v = fetch[k]
        kern[k]["fetch_bytes_per_launch"] = v / n * 1024 * 2      # gfx950 correction (x2)
        kern[k]["fetch_size_raw_kib_per_launch"] = v / n
    if k in write:
        n, v = write[k]
        kern[k]["write_bytes_per_launch"] = v / n * 1024
    if k in fetch_b:
        n, v = fetch_b[k]
        kern[k]["batched_launches"] = n
        kern[k]["fetch_bytes_per_batched_launch"] = v / n * 1024 * 2
    if k in write_b:
        n, v = write_b[k]
        kern[k]["write_bytes_per_batched_launch"] = v / n * 1024


def pmc_all(sub):
    """every counter of one pass, summed over the dispatches of a kernel (+ its total duration)"""
    try:
        f = newest(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    except ValueError:
        return {}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        did = r.get("Dispatch_Id", r.get("Dispatch_ID", ""))
        if did not in seen[k]:
            seen[k].add(did)
            agg[k]["_dispatches"] += 1
            agg[k]["_duration_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            agg[k]["_workgroups"] += float(r.get("Grid_Size", 0)) / max(float(r.get("Workgroup_Size", 1)), 1)
    return agg


sq = {}
for sub in ("sqa", "sqb"):
    for k, v in pmc_all(sub).items():
        sq.setdefault(k, {}).update({(c if not c.startswith("_") else f"{sub}{c}"): x for c, x in v.items()})
# L2 pass (round 5): hit rate and fabric read requests per kernel, batched launches only
try:
    tcc = {c: pmc("tcc", c, BATCHED_WGS) for c in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_EA0_RDREQ_sum")}
    for k in kern:
        if k in tcc["TCC_HIT_sum"] and k in tcc["TCC_MISS_sum"]:
            n, hit = tcc["TCC_HIT_sum"][k]
            _, miss = tcc["TCC_MISS_sum"][k]
            kern[k]["l2"] = {"batched_launches": n, "hit_per_launch": hit / n, "miss_per_launch": miss / n,
                             "hit_rate": hit / (hit + miss) if (hit + miss) else None,
                             "req_per_launch": tcc["TCC_REQ_sum"].get(k, [1, 0])[1] / n,
                             "fabric_rdreq_per_launch": tcc["TCC_EA0_RDREQ_sum"].get(k, [1, 0])[1] / n}
except (ValueError, KeyError, OSError):
    pass
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
# the same comparison on AVERAGES (what the bench line's roofline.launch_ms is): rocprofv3's per-kernel average
# mixes in the single-find launches of the untimed section (a few ciphertexts, ~3 ms each), so average over the
# dispatches of the batched steps only (>= 512 workgroups, as bench.py's timed launches are)
try:
    trace = newest(os.path.join(src, "stats", "*", "*_kernel_trace.csv"))
    # the dispatch sizes of a batched step: every level of one find x the strings of a step, cut into chunks of 16 384 (engine.hip);
    # the untimed sections of the command (one find in its log-depth form, the concurrent callers) launch other sizes
    sizes = None
    if bench_line and "config" in bench_line and bench_line["config"].get("pbs_per_level_of_one_find"):
        sizes = set()
        for lv in bench_line["config"]["pbs_per_level_of_one_find"]:
            c = lv * bench_line["config"]["strings_per_step_per_rank"]
            while c > 0:
                sizes.add(min(c, 16384))
                c -= min(c, 16384)
    def pbs_of(r):   # bootstraps of a dispatch: workgroups x bootstraps per workgroup (an odd launch rounds up: accept both)
        wg = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
        b = boots_of(r["Kernel_Name"])
        return [wg * b - k for k in range(b)]
    durs = [(float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6 for r in csv.DictReader(open(trace))
            if "blind_rotate" in r["Kernel_Name"] and max(pbs_of(r)) >= 512
            and (sizes is None or any(c in sizes for c in pbs_of(r)))]
    if check is not None and durs and bench_line:
        check["rocprof_avg_ms_of_batched_launches"] = round(sum(durs) / len(durs), 4)
        check["rocprof_batched_launches"] = len(durs)
        check["bench_launch_ms"] = bench_line["roofline"]["launch_ms"]
        check["avg_ratio"] = round(check["rocprof_avg_ms_of_batched_launches"] / bench_line["roofline"]["launch_ms"], 4)
except (ValueError, KeyError):
    pass
extra = sys.argv[2] if len(sys.argv) > 2 else ""      # e.g. "--params multibit2"
out = {"tag": tag, "command": f"python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline {extra}".strip() + " (under rocprofv3)",
       "duration_cross_check": check,
       "kernels": kern, "bench_line_under_profiler": bench_line}
json.dump(out, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
br = sorted((k for k in kern if "blind_rotate" in k), key=lambda k: -kern[k]["total_ms"])
if br:
    k = kern[br[0]]
    traffic = {"source": f"profiles/{tag}_summary.json", "kernel": br[0],
               "blind_rotate_hbm_bytes_per_batched_launch": round(k.get("fetch_bytes_per_batched_launch", 0) + k.get("write_bytes_per_batched_launch", 0)),
               "batched_launches": k.get("batched_launches"),
               "blind_rotate_hbm_bytes_per_launch_all": round(k.get("fetch_bytes_per_launch", 0) + k.get("write_bytes_per_launch", 0)),
               "note": "FETCH_SIZE x2 (gfx950) + WRITE_SIZE, KiB -> bytes; `per_batched_launch` averages the dispatches of >= 512 workgroups "
                       "(the batched steps bench.py times: same basis as its launch_ms), `per_launch_all` every dispatch of the run "
                       "(incl. the untimed single-find launches of 2-7 workgroups)"}
    params = extra.split()[-1] if "--params" in extra else "classic"
    name = "pmc_traffic.json" if params == "classic" else f"pmc_traffic_{params}.json"
    json.dump(traffic, open(os.path.join(dst, name), "w"), indent=1)
# ---- compute roofline inputs: SQ counters of the dominant kernel ----
rates_src = os.path.join(src, "issue_rates.jsonl")
rates = []
if os.path.exists(rates_src) and os.path.getsize(rates_src):
    shutil.copy(rates_src, os.path.join(dst, f"{tag}_issue_rates.jsonl"))
    rates = [json.loads(l) for l in open(rates_src) if l.strip().startswith("{")]
if br and br[0] in sq:
    c = sq[br[0]]
    # CMUX iterations (classic 2_2 set) / group steps (multi-bit: n / g = 764 / 2, 765 / 3) per PBS
    n_iter = 742 if "multibit" not in extra else (382 if extra.split()[-1] == "multibit2" else 255)
    wgs = c.get("sqa_workgroups", 0)                      # one workgroup = one PBS: 4 waves (throughput build), 2 (wave-per-polynomial build), 8 (latency builds)
    wpw = 2 if "_wpp_" in br[0] else (8 if ("_lat_" in br[0] or "_seq_" in br[0]) else 4)
    boots = boots_of(br[0])
    wgs = wgs * boots                                     # from here on: bootstraps, not workgroups
    wave_iters = wgs * wpw * (n_iter or 1)
    f64 = sum(c.get(x, 0) for x in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64"))
    wgs_b = (c.get("sqb_workgroups", 0) * boots) or wgs
    dur_a, dur_b = c.get("sqa_duration_ns", 0) * 1e-9, c.get("sqb_duration_ns", 0) * 1e-9
    clock = c.get("GRBM_GUI_ACTIVE", 0) / 8 / dur_a if dur_a else None     # MI355X_MICROARCH.md, DVFS give-back
    def rate(instr, wps=2):
        for r in rates:
            if r["instr"] == instr and r["waves_per_simd"] == wps:
                return r["cycles_from_wall_at_measured_clock"]
        return None
    c_f64 = rate("v_fma_f64")
    mix = rate("mix 5 fma_f64 : 3 int")
    c_other = (8 * mix - 5 * c_f64) / 3 if (mix and c_f64) else None
    comp = {
        "source": f"profiles/{tag}_summary.json (rocprofv3 --pmc passes sqa / sqb of the bench command), profiles/{tag}_issue_rates.jsonl",
        "kernel": br[0], "pbs_profiled": wgs, "cmux_iterations_per_pbs": n_iter,
        "valu_per_wave_iter": c.get("SQ_INSTS_VALU", 0) / wave_iters if wave_iters else None,
        "waves_per_pbs": wpw,
        "f64_per_wave_iter": f64 / (wgs_b * wpw * (n_iter or 1)) if wgs_b else None,
        "lds_instr_per_wave_iter": c.get("SQ_INSTS_LDS", 0) / (wgs_b * wpw * (n_iter or 1)) if wgs_b else None,
        "clock_ghz_under_load": clock / 1e9 if clock else None,
        "issue_cycles_f64_2waves": c_f64, "issue_cycles_other_next_to_f64": c_other,
        "valu_busy_frac": c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (c.get("GRBM_GUI_ACTIVE", 0) / 8 * 256 * 4) if c.get("GRBM_GUI_ACTIVE") else None,
        "lds_array_busy_frac": c.get("SQ_LDS_IDX_ACTIVE", 0) / (dur_b * clock * 256) if (dur_b and clock) else None,
        "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT", 0),
        "wave_time_split": {"active_any": c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"] * (dur_a / dur_b if dur_b else 1) if c.get("SQ_WAVE_CYCLES") else None,
                            "wait_any": c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None,
                            "wait_inst_any": c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None,
                            "wait_inst_lds": c.get("SQ_WAIT_INST_LDS", 0) / c["SQ_WAVE_CYCLES"] * (dur_a / dur_b if dur_b else 1) if c.get("SQ_WAVE_CYCLES") else None},
        "raw": {k2: v2 for k2, v2 in c.items()},
        "note": "SQ_* ACTIVE / WAIT / WAVE counters are quad-cycles; SQ_INSTS_* are wave-instructions; one workgroup = one PBS = waves_per_pbs waves x cmux_iterations",
    }
    if "multibit" not in extra:
        # Minimum number of f64 FMA / ADD / MUL instructions that produce the canonical arithmetic's bits, counted per
        # CMUX of the 2_2 set (N = 2048, M = 1024 points, k = 1, one level) and divided by the 256 lanes of a workgroup:
        #   2 forward transforms   9 stages x 512 butterflies x 6  +  the lane-pair stage 512 x 8                 = 2 x 31 744
        #   external product       2 outputs x 1024 points x 8                                                    =     16 384
        #   2 inverse transforms   first stage 512 x 4 adds + 9 x 512 x 6, minus 2 per butterfly whose twiddle is
        #                          exactly 1 or -i (1022 of them: s costs 2 instead of 4, d = fma(2, u, -s) stays)  = 2 x 27 652
        #   2 untwists             1024 points x 4 (1 / M and 2^-64 ride in the table), twist[0] = 1 skipped      = 2 x  4 092
        #   2 x 2048 roundings     6 per coefficient (y - rndne(y) form of 4.4)                                   = 2 x 12 288
        M_pts = 1024
        fwd = 9 * (M_pts // 2) * 6 + (M_pts // 2) * 8
        inv = (M_pts // 2) * 4 + 9 * (M_pts // 2) * 6 - 2 * (M_pts - 2)
        comp["f64_min_per_wave_iter"] = (2 * fwd + 2 * M_pts * 8 + 2 * inv + 2 * (M_pts * 4 - 4) + 2 * 2 * M_pts * 6) / (64 * wpw)
        comp["f64_min_note"] = ("minimum f64 FMA / ADD / MUL instructions per lane and CMUX that reproduce DESIGN.md section 4 bit for bit "
                                "(derivation in scripts/summarize_profile.py); measured f64_per_wave_iter equals it: the arithmetic is at its floor, "
                                "what is left to remove are the non-f64 VALU slots")
    if comp["valu_per_wave_iter"] and c_f64 and c_other and clock and n_iter:
        other = comp["valu_per_wave_iter"] - comp["f64_per_wave_iter"]
        cyc = comp["f64_per_wave_iter"] * c_f64 + other * c_other
        cyc = cyc * wpw / 4      # a CMUX occupies waves_per_pbs of the CU's four SIMDs for `cyc` cycles
        comp["issue_cycles_per_pbs_iteration_per_cu"] = cyc
        comp["bound_pbs_per_s"] = 256 * clock / (n_iter * cyc)
    # second pipe (round 3): the LDS.  Static instruction mix of the kernel's loop body (llvm-objdump of the
    # shipped code object) priced with the LDS-pipe cycles per wave-instruction of
    # MI355X_MICROARCH.md (ds_write_b128 13 -- the VGPR -> LDS transfer, not the array --, ds_read_b128 4,
    # ds_read_b64 2), plus the measured bank-conflict cycles; 4 waves per PBS share the CU's one LDS pipe.
    # scripts/micro/lds_valu_overlap.hip confirms the prices on this chip (8 stores + 8 loads by 8 waves:
    # 1021 cycles against 1088 predicted) and that a barrier-separated phase pays VALU + LDS, not max.
    if "multibit" not in extra and clock and wave_iters:
        mix = {"ds_write_b128": 48, "ds_read_b128": 58, "ds_read_b64": 16}   # 40 exchange + 18 twiddle reads (73 before the quarter-turn twiddles were derived, round 3)
        if wpw == 2:   # wave-per-polynomial build: 80 exchange / hand-over stores + 16 ds_write2st64_b64 of the image (priced as b128); 64 exchange + 32 product + 19 table loads; 32 rotated reads
            mix = {"ds_write_b128": 96, "ds_read_b128": 115, "ds_read_b64": 32}
        conflicts = c.get("SQ_LDS_BANK_CONFLICT", 0) / wave_iters
        lds_cyc = mix["ds_write_b128"] * 13 + mix["ds_read_b128"] * 4 + mix["ds_read_b64"] * 2 + conflicts
        comp["lds_pipe"] = {"instr_per_wave_iter_static": mix, "instr_per_wave_iter_measured": comp["lds_instr_per_wave_iter"],
                            "bank_conflict_cycles_per_wave_iter": conflicts,
                            "lds_pipe_cycles_per_wave_iter": lds_cyc,
                            "lds_pipe_cycles_per_pbs_iteration_per_cu": wpw * lds_cyc,
                            "bound_pbs_per_s": 256 * clock / (n_iter * wpw * lds_cyc)}
        if comp.get("issue_cycles_per_pbs_iteration_per_cu"):
            comp["serial_sum_pbs_per_s"] = 256 * clock / (n_iter * (comp["issue_cycles_per_pbs_iteration_per_cu"] + wpw * lds_cyc))
    json.dump(comp, open(os.path.join(dst, "compute_roofline.json" if "multibit" not in extra else f"compute_roofline_{extra.split()[-1]}.json"), "w"), indent=1)
    out["compute_roofline"] = comp
    json.dump(out, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
for k, v in kern.items():
    print(f"{k[:70]:70s} calls={v['calls']:5d} avg={v['avg_ms']:9.4f} ms  fetch/launch={v.get('fetch_bytes_per_launch', 0)/1e6:10.2f} MB")
