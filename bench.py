#!/usr/bin/env python3
"""bench.py -- headline benchmark (BASELINE.json): PBS/s and wall-clock of FheString::find on a
64-char encrypted input with a 4-char encrypted pattern (config 2), default shortint parameters.

  python bench.py --gpus N --steps K --warmup W

One "step" = one complete `find` (all of its bootstraps, 69 dependent levels) on inputs that are
already resident in HBM.  With N > 1 (launched by torch.distributed.run, one process per GPU)
every rank runs the same workload on its own encrypted string: independent strings shard with no
data-path collective, so scaling is "weak" and `value` is the PBS all ranks executed per second.

The JSON line also carries
  roofline      the dominant kernel (blind rotate): algorithmic bytes per launch / HIP-event
                launch duration, against the 8 TB/s HBM peak,
  cpu_baseline  the CPU oracle (a port of the same algorithm, NOT tfhe-rs) timed on this box's
                host cores on the same `find` program (rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per PBS (SURVEY.md 8d / BASELINE.md section 2), split per kernel:
KS_BYTES = 60_866_560 + 16_392                 # keyswitch key + LWE in
BR_BYTES = 48_627_712 + 32_768 + 16_392        # Fourier BSK + LUT accumulator + LWE out
PBS_BYTES = KS_BYTES + BR_BYTES                # 109,559,824
HBM_PEAK_GBS = 8000.0


def north_star_inputs(rank=0):
    import numpy as np
    rng = np.random.default_rng(0xF4E50002 + rank)
    s = "".join(chr(c) for c in rng.integers(0x20, 0x7F, size=64))
    return s, s[37:41]


def cpu_baseline(lib_params, text, pat, budget_s=30.0):
    """Times the oracle on the same compiled `find` program, all host cores."""
    import numpy as np
    from tests.helpers import World
    from tests.progs import Prog, num_blocks_usize, string_blocks, decode_result
    cores = os.cpu_count() or 1
    w = World("default", 255, 0xF4E50000)
    p = Prog(w.p, "find", [65, 5], num_blocks_usize(255))
    cts = w.encrypt(np.concatenate([string_blocks(text), string_blocks(pat)]))
    t0 = time.time()
    outs, _ = p.run_oracle(w, cts, threads=cores)
    dt = time.time() - t0
    ok = decode_result("find", w.decrypt(outs), num_blocks_usize(255)) == text.find(pat)
    return {"value": round(p.n_pbs / dt, 2), "unit": "PBS/s", "cores": cores, "kind": "port",
            "sample": f"one full find (64-char input, 4-char pattern): {p.n_pbs} PBS in {p.n_levels} levels, "
                      f"{dt:.1f} s wall on {cores} threads (OpenMP, one PBS per thread), result_ok={ok}",
            "wall_s": round(dt, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=0, help="blind-rotate kernel variant (0 = auto)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    dist = None
    if world_size > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_mod

    from fhe_string_amd import build
    if rank == 0:
        build.build_product(verbose=False)
        build.build_oracle(verbose=False)
    if dist:
        dist.barrier()
    from fhe_string_amd import api, _lib

    lib = _lib.load()
    params = api.default_params()
    ck = api.ClientKey(params, 255, seed=0xF4E50000)
    sk = api.ServerKey(ck, device=local_rank)
    _lib.check(lib.fhes_set_kernel_variant(sk.ctx, args.variant))
    text, pat = north_star_inputs(rank)
    a = api.FheString.new(ck, text)
    b = api.FheString.new(ck, pat)

    def barrier():
        sk.sync()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        sk.sync()

    result = None
    for _ in range(max(args.warmup, 0)):
        result = a.find(sk, b)      # uploads inputs on first use, compiles and caches the program
    barrier()
    sk.reset_stats()
    _lib.check(lib.fhes_set_timing(sk.ctx, 1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = a.find(sk, b)
    barrier()
    dt = time.perf_counter() - t0
    _lib.check(lib.fhes_set_timing(sk.ctx, 0))
    st = sk.stats()
    ok = result.decrypt(ck) == text.find(pat)

    dt_max = dt
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_max = float(t.item())

    if rank == 0:
        pbs_per_find = st.pbs_count // max(args.steps, 1)
        total_pbs = st.pbs_count * world_size
        launches = max(st.timed_launches, 1)
        br_ms = st.total_br_ms / launches
        ks_ms = st.total_ks_ms / launches
        units = st.pbs_count / launches
        br_achieved = (BR_BYTES * units) / (br_ms * 1e-3) / 1e9 if br_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("blind_rotate_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "PBS/s for FheString::find (64-char input, 4-char pattern)",
            "value": round(total_pbs / dt_max, 2),
            "unit": "PBS/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt_max / max(args.steps, 1) * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 torus (f64 FFT inside the external product)",
            "data": "synthetic",
            "config": {"workload": "FheString::find, 64-char unpadded input, 4-char encrypted pattern, "
                                   "PARAM_MESSAGE_2_CARRY_2_KS_PBS, l=255 (BASELINE config 2)",
                       "pbs_per_find": int(pbs_per_find), "levels": 69,
                       "strings_per_rank": 1, "parallelism": f"{world_size} independent strings"},
            "result_ok": bool(ok),
            "find_wall_ms": round(dt_max / max(args.steps, 1) * 1e3, 3),
            "roofline": {"bound": "hbm", "kernel": "blind_rotate", "achieved": round(br_achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(br_achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic,
                         "bytes_per_pbs": BR_BYTES, "pbs_per_launch": round(units, 2),
                         "launch_ms": round(br_ms, 4), "launches": int(st.timed_launches),
                         "note": "algorithmic bytes (key reuse across a batch makes HBM traffic smaller)"},
            "roofline_pbs": {"bound": "hbm", "kernel": "keyswitch+blind_rotate",
                             "achieved": round(PBS_BYTES * units / ((br_ms + ks_ms) * 1e-3) / 1e9, 1) if br_ms + ks_ms > 0 else 0.0,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "bytes_per_pbs": PBS_BYTES,
                             "keyswitch_launch_ms": round(ks_ms, 4)},
        }
        line["roofline_pbs"]["frac"] = round(line["roofline_pbs"]["achieved"] / HBM_PEAK_GBS, 4)
        if world_size == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(params, text, pat)
        print(json.dumps(line), flush=True)

    sk.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
