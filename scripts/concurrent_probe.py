"""Developer probe: wall-clock of N threads each running one config-2 find on one context."""
import sys, threading, time
import numpy as np
sys.path.insert(0, ".")
from fhe_string_amd._lib import check
from tests.helpers import ptr, world
from tests.progs import Prog, num_blocks_usize, string_blocks

w = world("default")
bu = num_blocks_usize(255)
p = Prog(w.p, "find", [65, 5], bu)
ctx = w.ctx()
NMAX = 16
handles = []
for k in range(NMAX):
    rng = np.random.default_rng(100 + k)
    s = "".join(chr(c) for c in rng.integers(0x20, 0x7F, size=64))
    cts = w.encrypt(np.concatenate([string_blocks(s), string_blocks(s[5:9])]))
    h = check(w.lib.fhes_arena_alloc(ctx, len(cts)))
    check(w.lib.fhes_arena_upload(ctx, h, len(cts), ptr(cts)))
    handles.append(np.arange(h, h + len(cts), dtype=np.int32))
for n in [1, 1, 2, 4, 8, 16, 8]:
    outs = [np.zeros(p.n_outputs, dtype=np.int32) for _ in range(n)]
    gate = threading.Barrier(n + 1)
    tq = [0.0] * n
    def worker(k):
        gate.wait()
        t = time.perf_counter()
        check(w.lib.fhes_program_run(ctx, p.h, ptr(handles[k]), ptr(outs[k])))
        tq[k] = time.perf_counter() - t
    th = [threading.Thread(target=worker, args=(k,)) for k in range(n)]
    for t in th: t.start()
    gate.wait()
    t0 = time.perf_counter()
    for t in th: t.join()
    check(w.lib.fhes_sync(ctx))
    dt = time.perf_counter() - t0
    for o in outs:
        check(w.lib.fhes_arena_free(ctx, int(o[0]), p.n_outputs))
    print(f"threads={n:2d} wall={dt*1e3:8.1f} ms  per-call host time min/max = {min(tq)*1e3:.1f}/{max(tq)*1e3:.1f} ms", flush=True)
# reference point: the same 8 finds as ONE batched run (levels merged into common launches)
for n in [1, 8, 16]:
    ins = np.concatenate(handles[:n])
    out = np.zeros(n * p.n_outputs, dtype=np.int32)
    check(w.lib.fhes_sync(ctx))
    t0 = time.perf_counter()
    check(w.lib.fhes_program_run_batch(ctx, p.h, n, ptr(ins), ptr(out)))
    check(w.lib.fhes_sync(ctx))
    print(f"batched n_inst={n:2d} wall={(time.perf_counter()-t0)*1e3:8.1f} ms", flush=True)
    check(w.lib.fhes_arena_free(ctx, int(out[0]), len(out)))
